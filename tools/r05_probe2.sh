#!/bin/bash
O=gpurun_out/r05_probe2
mkdir -p $O
timeout 900 python -m pytest tests/test_lazy_adaptive.py tests/test_host_api.py tests/test_flagship_golden.py tests/test_bench_line.py -m gpu -q -x -p no:cacheprovider -s 2>&1 | tail -70 > $O/gpu_tests_quick.log; tail -25 $O/gpu_tests_quick.log
timeout 300 python tools/lazy_sweep.py > $O/lazy_sweep.json 2> $O/lazy_sweep.err; cat $O/lazy_sweep.err | tail -12
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; cut -c1-600 $O/bench_driver_like.json
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_200.json 2> $O/bench_200.err
python - <<'PY'
import json
for f in ("bench_driver_like", "bench_200"):
    try:
        d = json.load(open(f"gpurun_out/r05_probe2/{f}.json"))
        print(f, "ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], "backpressure", d.get("host_backpressure_ms_per_step"), "gpu", d["gpu_ms_per_step_events"],
              {k: round(v["avg_launch_us"], 1) for k, v in d["roofline"]["per_kernel"].items()}, d.get("one_step_parity"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 python tools/host_profile.py --steps 300 > $O/host_profile_single.txt 2>&1; head -30 $O/host_profile_single.txt
