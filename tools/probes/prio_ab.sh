mkdir -p gpurun_out/r06_prio
for rep in 1 2 3; do for v in base T N; do
  L=morl-baselines_amd/lib/libmorl_hip.so; [ $v != base ] && L=tools/probes/libmorl_prio_$v.so
  MORL_HIP_LIB=$L timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record > gpurun_out/r06_prio/${v}_$rep.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/r06_prio/${v}_$rep.json').read()); print('$v', round(j['ms_per_step'],4), {k: round(x['avg_launch_us'], 1) for k, x in j['roofline']['per_kernel'].items()})"
done; done
