#!/bin/bash
O=gpurun_out/r05_probe8
mkdir -p $O
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  MORL_PROBE_FIT_GRID=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_fit$v.json 2>/dev/null
  (cd /tmp && export TMPDIR=/tmp && MORL_PROBE_FIT_GRID=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fit$v -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1)
  for f in $(find $O/prof_fit$v -name "*kernel_stats.csv"); do echo == fit $v; grep "chain4\|chain_bf\|dw_bf" $f | cut -c1-110; done
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
for v in (1, 0):
    d = json.load(open(f"gpurun_out/r05_probe8/bench_fit{v}.json")); print("fit", v, d["ms_per_step"], d["lazy_target_rows_last_step"])
PY
timeout 600 python -m pytest tests/test_lazy_adaptive.py tests/test_flagship_golden.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
