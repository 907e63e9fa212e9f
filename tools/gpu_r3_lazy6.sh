set -x
O=gpurun_out/lazy6
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
cd $R
python tools/trace_phases.py $(find $O/prof -name "*kernel_trace.csv") > $O/prof.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_flagship_golden.py -m gpu -q -p no:cacheprovider -k "lazy or golden" 2>&1 | tail -5 > $O/tests.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_lazy.json 2>$O/bench_lazy.err
