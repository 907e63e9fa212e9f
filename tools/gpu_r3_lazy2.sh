set -x
O=gpurun_out/lazy2
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py tests/test_train_traces.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6 > $O/tests.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_lazy_1.json 2>$O/bench_lazy_1.err
MORL_LAZY_TARGETS=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_eager_1.json 2>/dev/null
timeout 300 python tools/diag_single_pass.py > $O/single_pass.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sp -- python $R/tools/diag_single_pass.py 16384 > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
