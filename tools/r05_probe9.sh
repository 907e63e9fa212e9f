#!/bin/bash
O=gpurun_out/r05_probe9
mkdir -p $O
timeout 1200 python -m pytest tests/test_distributed.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
B="--no-cpu-baseline --no-ramp-record --steps 100 --warmup 20"
for n in 2 4 8; do for ax in batch weights; do
  timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > $O/emu${n}_${ax}.json 2>/dev/null
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_probe9/emu*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], "ms %.4f host %.4f back %.4f" % (d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("host_backpressure_ms_per_step", 0)))
PY
