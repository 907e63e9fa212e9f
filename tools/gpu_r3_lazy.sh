# lazy target evaluation: gpu parity tests that touch it, A/B bench lines, kernel stats
set -x
O=gpurun_out/lazy
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py tests/test_train_traces.py tests/test_host_api.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6 > $O/tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_lazy_$i.json 2>$O/bench_lazy_$i.err
MORL_LAZY_TARGETS=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_eager_$i.json 2>/dev/null
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lazy_driver.json 2>/dev/null
MORL_LAZY_TARGETS=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_eager_driver.json 2>/dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -3 $O/tests.log
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('lazy_target_rows_last_step'), d['roofline']['frac'], {k:v.get('avg_launch_us') for k,v in d['roofline'].get('per_kernel',{}).items()})
"; done
find $O -name "*kernel_stats.csv" | head -1 | xargs head -14
