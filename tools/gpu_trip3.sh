set -x
R=$PWD
O=gpurun_out/t3
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py tests/test_train_traces.py -m gpu -x -q 2>&1 | tail -5
for dw in 1 3; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dw-mode $dw > $O/bench_dw$dw.json 2> $O/bench_dw$dw.err
  python - <<PY
import json
d=json.load(open("$O/bench_dw$dw.json"))
print("dw$dw: ms/step %.4f host_enq %.4f chain avg us %.1f frac %.3f loss %.6f"%(d["ms_per_step"], d["host_enqueue_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["last_loss"]))
PY
done
for j in 384 448 640 768 1024; do
  MORL_DW_JOBS=$j timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_dwjobs$j.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_dwjobs$j.json')); print('dw jobs $j: ms/step %.4f'%d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-150
cd $R
