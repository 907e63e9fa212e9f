set -x
O=gpurun_out/lazy3
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py tests/test_train_traces.py tests/test_chain_tilings.py -m gpu -q -p no:cacheprovider -x -k "lazy or golden or trace" 2>&1 | tail -6 > $O/tests.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record"
timeout 300 $B > $O/bench_mix.json 2>$O/bench_mix.err
MORL_LAZY_MIX=0 timeout 300 $B > $O/bench_nomix.json 2>/dev/null
MORL_LAZY_TARGETS=0 timeout 300 $B > $O/bench_eager.json 2>/dev/null
MORL_LAZY_MIX_SLOTS=512 timeout 300 $B > $O/bench_mix_s512.json 2>/dev/null
MORL_LAZY_MIX_SLOTS=384 timeout 300 $B > $O/bench_mix_s384.json 2>/dev/null
timeout 300 $B > $O/bench_mix_2.json 2>/dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
