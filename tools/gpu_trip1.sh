# GPU trip 1 (round 2): GPU test suite, then A/B of the chain kernel generations inside bench.py, then a kernel-stats profile.
set -x
R=$PWD
O=gpurun_out/t1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
for cfg in "1 0" "2 0" "2 1"; do
  set -- $cfg
  MORL_CHAIN_GEN=$1 MORL_CHAIN_SCHED=$2 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_gen$1_s$2.json 2> $O/bench_gen$1_s$2.err
  python - <<PY
import json
d=json.load(open("$O/bench_gen$1_s$2.json"))
print("gen$1 sched$2: ms/step %.4f host_enq %.4f chain avg us %.1f frac %.3f launches %d"%(d["ms_per_step"], d["host_enqueue_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["launches_timed"]))
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err
cut -c1-600 $O/bench_driver_like.json
timeout 300 python bench.py --gpus 1 --force-shard --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_force_shard.json 2> $O/bench_force_shard.err; cut -c1-300 $O/bench_force_shard.json
cd /tmp && export TMPDIR=/tmp
for cfg in "2 1" "2 0"; do
  set -- $cfg
  MORL_CHAIN_GEN=$1 MORL_CHAIN_SCHED=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gen$1_s$2 -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $R/$O/prof_gen$1_s$2 -name "*kernel_stats.csv" | head -1); echo $f; head -14 $f
done
cd $R
