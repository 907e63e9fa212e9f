#!/bin/bash
# A/B of the weight-gradient kernel's producer look-ahead (old = the commit before)
O=gpurun_out/r05_probe17
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
for i in 1 2 3; do for v in old new; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/probe_old/libmorl_hip.so; fi
  MORL_HIP_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 200 --warmup 30 > $O/bench_${v}_$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); pk=d['roofline'].get('per_kernel',{}); print('$v', d['ms_per_step'], 'dW us', pk.get('dw',{}).get('avg_launch_us'))"
done; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench -o p -- python $R/bench.py --no-cpu-baseline --no-ramp-record --steps 80 --warmup 20 > /dev/null 2>&1)
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob("gpurun_out/r05_probe17/prof_*/**/*.db", recursive=True)):
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
    for r in c.execute(q):
        if "morl" in r[0]: print("   %-60s %6d avg %.0f min %.0f" % (r[0][:60], r[1], r[2], r[3]))
PY
timeout 900 python -m pytest tests/test_flagship_golden.py tests/test_kernels_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
