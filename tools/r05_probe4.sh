#!/bin/bash
O=gpurun_out/r05_probe4
mkdir -p $O
timeout 1500 python -m pytest tests/test_ln_chain.py tests/test_gpi_kernels_parity.py tests/test_ac_kernels_parity.py tests/test_gpi_agent.py tests/test_ac_agents.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > $O/gpu_tests_ln.log; tail -6 $O/gpu_tests_ln.log
for w in gpi gpipd; do
  timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/bench_ac_${w}.json 2>/dev/null
  MORL_AC_LN_CHAIN=0 timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/bench_ac_${w}_per_layer.json 2>/dev/null
done
timeout 300 python bench_ac.py --workload capql --no-cpu-baseline > $O/bench_ac_capql.json 2>/dev/null
timeout 300 python bench_ac.py --workload mosac --no-cpu-baseline > $O/bench_ac_mosac.json 2>/dev/null
timeout 300 python bench_ac.py --workload morld --pop 64 --no-cpu-baseline > $O/bench_ac_morld64.json 2>/dev/null
timeout 300 python bench_ac.py --workload morld --pop 64 --devices 2 --no-cpu-baseline > $O/bench_ac_morld64_2ctx.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_probe4/bench_ac_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], "ms", round(d["ms_per_step"], 4), d.get("per_loop", {}).get("ms_per_env_step"), d.get("per_loop", {}).get("host_enqueue_ms_per_env_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gpi -- python $R/bench_ac.py --workload gpi --steps 60 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gpipd -- python $R/bench_ac.py --workload gpipd --steps 60 --no-cpu-baseline > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for f in $(find $O/prof_gpi -name "*kernel_stats.csv"); do head -22 $f | cut -c1-150; done
