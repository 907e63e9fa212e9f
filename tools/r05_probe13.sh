#!/bin/bash
# A/B of the 16-row chain's look-ahead fix: actor-critic / GPI updates and a rank of an 8-rank step, old library against new
O=gpurun_out/r05_probe13
mkdir -p $O
R=$PWD
for v in old new; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/probe_old/libmorl_hip.so; fi
  for w in capql mosac gpipd gpi; do
    MORL_HIP_LIB=$L timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/ac_${w}_$v.json 2>/dev/null
    python -c "
import json; d=json.load(open('$O/ac_${w}_$v.json')); print('$v $w', d['ms_per_step'])"
  done
  MORL_EXACT_F32=1 MORL_HIP_LIB=$L timeout 200 python bench.py --gpus 1 --force-shard --emulate-world 8 --shard-axis batch --no-cpu-baseline --no-ramp-record --steps 100 --warmup 20 > $O/emu8_f32_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/emu8_f32_$v.json')); print('$v emu8 exact-f32 batch axis', d['ms_per_step'])"
done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
