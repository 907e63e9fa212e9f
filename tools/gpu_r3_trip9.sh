O=gpurun_out/r3t9; mkdir -p $O
for v in 0 1 0 1; do MORL_DW_NT=$v timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/nt_$v.json 2>/dev/null; python - <<PY
import json
j=json.load(open("$O/nt_$v.json")); print("DW_NT=$v", round(j["ms_per_step"],4), {k:round(v["avg_launch_us"],1) for k,v in j["roofline"]["per_kernel"].items()})
PY
done
