set -x
R=$PWD
O=gpurun_out/t12
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("ms/step %.4f host_enq %.4f chain avg us %.1f frac %.3f loss %.6f"%(d["ms_per_step"], d["host_enqueue_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["last_loss"]))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench20.json')); print('20-step: ms/step %.4f'%d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-130
cd $R
