#!/bin/bash
O=gpurun_out/r05_probe3
mkdir -p $O
timeout 1200 python -m pytest tests/test_distributed.py tests/test_lazy_adaptive.py tests/test_host_api.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30 > $O/gpu_tests_dist.log; tail -8 $O/gpu_tests_dist.log
B="--no-cpu-baseline --no-ramp-record --steps 100 --warmup 20"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/single.json 2>/dev/null
for n in 2 4 8; do
  for ax in batch weights; do
    timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > $O/emu${n}_${ax}.json 2>/dev/null
    MORL_BF_MIN_ROWS=4096 MORL_LAZY_MIN_ROWS=4096 timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > $O/t4096_emu${n}_${ax}.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in ["gpurun_out/r05_probe3/single.json"] + sorted(glob.glob("gpurun_out/r05_probe3/*emu*.json")):
    try:
        d = json.load(open(f))
        pk = d["roofline"]["per_kernel"]
        print(f.split("/")[-1], "ms %.4f host %.4f back %.4f gpu %.4f" % (d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("host_backpressure_ms_per_step", 0), d["gpu_ms_per_step_events"]),
              {k: round(v["avg_launch_us"], 1) for k, v in pk.items()}, d["dtype"][:8], "lazy", d.get("lazy_target_rows_last_step"), "execfrac", round(d["roofline"].get("whole_step_frac_executed", 0), 3))
    except Exception as e:
        print(f, "ERR", e)
PY
python tools/emulated_ceiling.py $O/single.json $O $O/emulated_ceiling.json > /dev/null
timeout 300 python tools/host_profile.py --steps 300 > $O/host_profile_single.txt 2>&1; head -12 $O/host_profile_single.txt
