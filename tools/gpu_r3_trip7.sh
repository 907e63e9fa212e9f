O=gpurun_out/r3t7; mkdir -p $O
for j in 384 448 480 544 640 768 1024; do MORL_DW_JOBS=$j timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/dw_$j.json 2>/dev/null; done
for s in 256 384 768 1024; do MORL_CHAIN_SLOTS=$s timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/slots_$s.json 2>/dev/null; done
for w in 4 16; do MORL_TD_WAVES=$w timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/tdw_$w.json 2>/dev/null; done
timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 100 > $O/base.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3t7/*.json")):
    try:
        j=json.load(open(f)); print(f.split('/')[-1].ljust(16), round(j["ms_per_step"],4), {k:round(v["avg_launch_us"],1) for k,v in j["roofline"]["per_kernel"].items()})
    except Exception as e: print(f, "ERR")
PY
