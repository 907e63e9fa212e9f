O=gpurun_out/t10; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "0 1" "1 1" "1 0"; do
set -- $cfg
for w in capql mosac gpipd gpi; do
  MORL_AC_CHAIN=$1 MORL_CHAIN16=$2 timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/ac_${w}_c$1_s$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/ac_${w}_c$1_s$2.json')); print('ac_chain=$1 chain16=$2 $w: ms/update %.4f'%d['ms_per_step'])"
done
done
MORL_AC_CHAIN=1 timeout 300 python bench_ac.py --workload morld --pop 64 --no-cpu-baseline > $O/ac_morld64.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/ac_morld64.json')); print('morld64 chain: ms/update %.4f'%d['ms_per_step'])"
# a shard-sized Envelope step on one GPU (what a rank of an 8-GPU strong-scaled job computes): W = 8 of 64
for s in 1 0; do
MORL_CHAIN16=$s timeout 300 python bench.py --weights 8 --steps 200 --warmup 20 --no-cpu-baseline > $O/env_w8_s$s.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/env_w8_s$s.json')); print('envelope B=256 x W=8 chain16=$s: ms/step %.4f'%d['ms_per_step'])"
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print('envelope: ms/step %.4f'%(d['ms_per_step']))"
