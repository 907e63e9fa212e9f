mkdir -p gpurun_out/r06_pwfwd
timeout 200 python -m pytest tests/test_chain_tilings.py -m gpu -q -x -p no:cacheprovider -k "producer_wave_in_the_forward" 2>&1 | tail -3
for rep in 1 2 3; do for d in 0 1; do
  MORL_BF_PW_FWD=$d timeout 120 python bench.py --weights 32 --steps 300 --warmup 20 --no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record > gpurun_out/r06_pwfwd/w32_${d}_$rep.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/r06_pwfwd/w32_${d}_$rep.json').read()); print('w32 fwd_pw=$d', round(j['ms_per_step'],4), {k: round(x['avg_launch_us'], 1) for k, x in j['roofline']['per_kernel'].items()})"
  MORL_BF_PW_FWD=$d timeout 120 python bench.py --gpus 1 --force-shard --emulate-world 2 --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record > gpurun_out/r06_pwfwd/emu2_${d}_$rep.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/r06_pwfwd/emu2_${d}_$rep.json').read()); print('rank-of-2 fwd_pw=$d', round(j['ms_per_step'],4), {k: round(x['avg_launch_us'], 1) for k, x in j['roofline']['per_kernel'].items()})"
done; done
