O=gpurun_out/r3t6; mkdir -p $O; R=$PWD
timeout 200 python bench.py --force-shard --emulate-world 8 --shard-axis batch --no-cpu-baseline --steps 100 > $O/bench_emu8_batch.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_emu8 -- python $R/bench.py --force-shard --emulate-world 8 --shard-axis batch --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json,glob,csv
d=json.load(open("gpurun_out/r3t6/bench_emu8_batch.json")); print("emu8 batch ms", d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
f=glob.glob("gpurun_out/r3t6/prof_emu8/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), round(float(r['AverageNs'])/1e3,2))
PY
