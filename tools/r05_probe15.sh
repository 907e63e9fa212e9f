#!/bin/bash
O=gpurun_out/r05_probe15
mkdir -p $O
R=$PWD
for i in 1 2; do for v in old new; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/probe_old/libmorl_hip.so; fi
  MORL_HIP_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 200 --warmup 30 > $O/bench_${v}_$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('$v bench', d['ms_per_step'])"
done; done
for v in old new; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/probe_old/libmorl_hip.so; fi
  for w in capql mosac gpipd gpi; do
    MORL_HIP_LIB=$L timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/ac_${w}_$v.json 2>/dev/null
    python -c "
import json; d=json.load(open('$O/ac_${w}_$v.json')); print('$v $w', d['ms_per_step'])"
  done
done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
