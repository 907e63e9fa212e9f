O=gpurun_out/r3t12; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -p no:cacheprovider -k "shared_gpu" 2>&1 | tail -4
timeout 600 python bench.py --gpus 4 --shared-gpu --steps 20 --warmup 5 > $O/bench_shared_gpu_4.json 2> $O/err4.txt
timeout 600 python bench.py --gpus 2 --shared-gpu --steps 20 --warmup 5 > $O/bench_shared_gpu_2.json 2> $O/err2.txt
python - <<'PY'
import json
for n in (2,4):
    d=json.load(open(f"gpurun_out/r3t12/bench_shared_gpu_{n}.json")); print(n, d["ms_per_step"], d.get("collectives_alone",{}).get("ipc"))
PY
