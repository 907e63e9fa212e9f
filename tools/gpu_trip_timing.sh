# Event-record overhead of the roofline timing on a driver-shaped run; kernel breakdown of the shard-sized steps.
set -x
R=$PWD
O=gpurun_out/timing
mkdir -p $O
python -m pytest tests/test_host_api.py -m gpu -x -q -k timing 2>&1 | tail -2 > $O/test.log
for rep in 1 2; do
for m in 1 -1 0 4; do
  MORL_BENCH_TIMING=$m python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_t${m}_$rep.json 2>/dev/null
done
MORL_EV_FLAGS=0x20000000 MORL_BENCH_TIMING=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_t1_nofence_$rep.json 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_w8 -- python $R/bench.py --weights 8 --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_shard -- python $R/bench.py --force-shard --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $R
cat $O/test.log
for f in $O/drv_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['gpu_ms_per_step_events'], d['roofline']['launches_timed'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
done
