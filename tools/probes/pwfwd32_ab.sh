mkdir -p gpurun_out/r06_pwfwd32
timeout 200 python -m pytest tests/test_chain_tilings.py -m gpu -q -x -p no:cacheprovider -k "forward_launch_on_32" 2>&1 | tail -3
for rep in 1 2 3; do for d in 0 1; do
  MORL_BF_PW_FWD32=$d timeout 120 python bench.py --gpus 1 --force-shard --emulate-world 4 --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record > gpurun_out/r06_pwfwd32/emu4_${d}_$rep.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/r06_pwfwd32/emu4_${d}_$rep.json').read()); print('rank-of-4 fwd32_pw=$d', round(j['ms_per_step'],4), {k: round(x['avg_launch_us'], 1) for k, x in j['roofline']['per_kernel'].items()})"
  MORL_BF_PW_FWD32=$d timeout 120 python bench.py --batch 128 --weights 32 --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record > gpurun_out/r06_pwfwd32/b128w32_${d}_$rep.json 2>/dev/null
  python -c "import json; j=json.loads(open('gpurun_out/r06_pwfwd32/b128w32_${d}_$rep.json').read()); print('128x32 fwd32_pw=$d', round(j['ms_per_step'],4), {k: round(x['avg_launch_us'], 1) for k, x in j['roofline']['per_kernel'].items()})" 2>/dev/null
done; done
