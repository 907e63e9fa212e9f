set -x
O=gpurun_out/r3t3; mkdir -p $O
timeout 600 python -m pytest tests/test_chain_tilings.py -m gpu -q -k "td_stage" -p no:cacheprovider 2>&1 | tail -5 > $O/td_fused_test.log
timeout 200 python bench.py --no-cpu-baseline --no-ramp-record > $O/bench_fused.json 2> $O/bench_fused.err
MORL_TD_FUSED=0 timeout 200 python bench.py --no-cpu-baseline --no-ramp-record > $O/bench_unfused.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --no-ramp-record --steps 20 --warmup 5 > $O/bench_fused_20.json 2>/dev/null
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/td_fused_test.log; cut -c1-300 $O/bench_fused.json; echo; cut -c1-300 $O/bench_unfused.json
