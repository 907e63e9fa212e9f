#!/bin/bash
# round-5 probe 1: what the existing bf16 + lazy pipeline does at the row counts of a sharded job's rank (8 192 / 4 096 / 2 048 rows)
O=gpurun_out/r05_probe1
mkdir -p $O
B="--no-cpu-baseline --no-ramp-record --steps 100 --warmup 20"
for n in 2 4 8; do
  for ax in batch weights; do
    timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > $O/emu${n}_${ax}_default.json 2>/dev/null
    MORL_BF_MIN_ROWS=0 MORL_LAZY_MIN_ROWS=0 timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > $O/emu${n}_${ax}_bf_lazy.json 2>/dev/null
    MORL_BF_MIN_ROWS=0 timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > $O/emu${n}_${ax}_bf_eager.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_probe1/emu*.json")):
    try:
        d = json.load(open(f))
        pk = d["roofline"]["per_kernel"]
        print(f.split("/")[-1], "ms %.4f host %.4f gpu %.4f" % (d["ms_per_step"], d["host_enqueue_ms_per_step"], d["gpu_ms_per_step_events"]),
              {k: round(v["avg_launch_us"], 1) for k, v in pk.items()}, "bf", d.get("roofline", {}).get("kernel", "")[:14], "lazy", d.get("lazy_target_rows_last_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 python tools/host_profile.py --steps 300 > $O/host_profile_single.txt 2>&1; head -60 $O/host_profile_single.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MORL_BF_MIN_ROWS=0 MORL_LAZY_MIN_ROWS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_emu8_bf -- python $R/bench.py --gpus 1 --force-shard --emulate-world 8 --shard-axis batch --no-cpu-baseline --no-ramp-record --steps 80 --warmup 10 > /dev/null 2>&1
cd $R
find $O/prof_emu8_bf -name "*kernel_trace.csv" -delete; find $O/prof_emu8_bf -name "*agent_info.csv" -delete
for f in $(find $O/prof_emu8_bf -name "*kernel_stats.csv"); do head -16 $f | cut -c1-150; done
