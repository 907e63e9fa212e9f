set -x
O=gpurun_out/lazy4
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py tests/test_train_traces.py tests/test_chain_tilings.py -m gpu -q -p no:cacheprovider -k "lazy or eager or golden or trace" 2>&1 | tail -40 > $O/tests.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record"
timeout 300 $B > $O/bench_lazy.json 2>$O/bench_lazy.err
MORL_LAZY_WARM=0 timeout 300 $B > $O/bench_lazy_nowarm.json 2>/dev/null
MORL_LAZY_TARGETS=0 timeout 300 $B > $O/bench_eager.json 2>/dev/null
timeout 300 $B > $O/bench_lazy_2.json 2>/dev/null
timeout 300 $B --weights 64 > $O/bench_w64_lazy.json 2>/dev/null
MORL_LAZY_TARGETS=0 timeout 300 $B --weights 64 > $O/bench_w64_eager.json 2>/dev/null
timeout 300 $B --weights 24 > $O/bench_w24_lazy.json 2>/dev/null
MORL_LAZY_TARGETS=0 timeout 300 $B --weights 24 > $O/bench_w24_eager.json 2>/dev/null
E="python bench.py --gpus 1 --force-shard --emulate-world 8 --no-cpu-baseline"
timeout 300 $E > $O/bench_emu8_default.json 2>/dev/null
MORL_LAZY_MIN_ROWS=0 timeout 300 $E > $O/bench_emu8_lazy.json 2>/dev/null
E="python bench.py --gpus 1 --force-shard --emulate-world 2 --no-cpu-baseline"
timeout 300 $E > $O/bench_emu2_default.json 2>/dev/null
MORL_LAZY_MIN_ROWS=0 timeout 300 $E > $O/bench_emu2_lazy.json 2>/dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
