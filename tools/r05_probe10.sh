#!/bin/bash
O=gpurun_out/r05_probe10
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in new probe_old probe2 probe3; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/$v/libmorl_hip.so; fi
  echo "== $v"
  MORL_HIP_LIB=$L timeout 120 python tools/chain4_rows.py 2>&1 | tail -4
  (cd /tmp && MORL_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o p -- python $R/tools/chain4_rows.py > /dev/null 2>&1)
  python - <<PY
import csv, glob
for f in glob.glob("$R/$O/prof_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain4" in r["Name"]: print("   rocprof", r["Name"][:40], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
done
for v in new probe_old; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/$v/libmorl_hip.so; fi
  for i in 1 2; do
    MORL_HIP_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 200 --warmup 30 > $O/bench_${v}_$i.json 2>/dev/null
    python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('$v', d['ms_per_step'], d['config'].get('lazy_target_rows_last_step'))"
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench -o p -- python $R/bench.py --no-cpu-baseline --no-ramp-record --steps 80 --warmup 20 > /dev/null 2>&1)
python - <<PY
import csv, glob
for f in glob.glob("$R/$O/prof_bench/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("   bench", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
timeout 900 python -m pytest tests/test_chain_tilings.py tests/test_lazy_adaptive.py tests/test_flagship_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
