#!/bin/bash
O=gpurun_out/r05_probe11
mkdir -p $O
R=$PWD
echo "== phase stamps"
MORL_HIP_LIB=$R/morl-baselines_amd/lib/probe_prof/libmorl_hip.so timeout 120 python tools/chain4_rows.py 2>&1 | grep -v "^rows" | tee $O/c4_prof.txt | head -40
echo "== kernarg placement"
for k in 0 1; do
  for i in 1 2; do
    HIP_FORCE_DEV_KERNARG=$k timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 200 --warmup 30 > $O/bench_kernarg${k}_$i.json 2>/dev/null
    python -c "
import json; d=json.load(open('$O/bench_kernarg${k}_$i.json')); print('HIP_FORCE_DEV_KERNARG=$k', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
  done
done
