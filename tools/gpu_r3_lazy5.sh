set -x
O=gpurun_out/lazy5
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
MORL_CHAIN4=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_chain16 -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
cd $R
for d in prof prof_chain16; do python tools/trace_phases.py $(find $O/$d -name "*kernel_trace.csv") > $O/$d.txt; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py tests/test_train_traces.py tests/test_chain_tilings.py -m gpu -q -p no:cacheprovider -k "lazy or eager or golden or trace" 2>&1 | tail -5 > $O/tests.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record"
timeout 300 $B > $O/bench_lazy.json 2>$O/bench_lazy.err
MORL_LAZY_TARGETS=0 timeout 300 $B > $O/bench_eager.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lazy_driver.json 2>/dev/null
