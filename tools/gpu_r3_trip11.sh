O=gpurun_out/r3t11; mkdir -p $O
timeout 600 python bench.py --gpus 2 --shared-gpu --steps 20 --warmup 5 > $O/bench_shared_gpu_2.json 2> $O/err2.txt
timeout 600 python bench.py --gpus 4 --shared-gpu --steps 20 --warmup 5 > $O/bench_shared_gpu_4.json 2> $O/err4.txt
timeout 300 python bench.py --gpus 1 --force-shard --emulate-world 8 --no-cpu-baseline > $O/bench_emu8.json 2>/dev/null
python - <<'PY'
import json
for n in (2,4):
    try:
        d=json.load(open(f"gpurun_out/r3t11/bench_shared_gpu_{n}.json")); print(n, d["ms_per_step"], d.get("collectives_alone"))
    except Exception as e:
        print(n, "ERR", e); print(open(f"gpurun_out/r3t11/err{n}.txt").read()[-1500:])
d=json.load(open("gpurun_out/r3t11/bench_emu8.json")); print("emu8", {k:v["ms_per_step"] for k,v in d["strong_scaling_axes"].items()})
PY
