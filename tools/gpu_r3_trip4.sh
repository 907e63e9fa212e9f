O=gpurun_out/r3t4; mkdir -p $O
for d in 0 1 2 4 8 15 3 7; do MORL_TD_DBG=$d timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/b_$d.json 2>/dev/null; done
python - <<'PY'
import json
for d in (0,1,2,4,8,15,3,7):
    try:
        j=json.load(open(f"gpurun_out/r3t4/b_{d}.json")); print("dbg",d, round(j["ms_per_step"],4), {k:round(v["avg_launch_us"],1) for k,v in j["roofline"]["per_kernel"].items()})
    except Exception as e: print(d, e)
PY
