O=gpurun_out/r3t10; mkdir -p $O
for v in 0 1 0 1; do MORL_DW_LAYOUT1=$v timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/l1_$v.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/l1_$v.json")); print("DW_LAYOUT1=$v", round(j["ms_per_step"],4), {k:round(v["avg_launch_us"],1) for k,v in j["roofline"]["per_kernel"].items()}, j["last_loss"])
PY
done
