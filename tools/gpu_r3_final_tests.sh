set -x
O=gpurun_out/final_r3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -80 > $O/gpu_tests.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 300 python bench.py > $O/bench_per_on.json 2> $O/bench_per_on.err
tail -3 $O/gpu_tests.log; tail -2 $O/smoke.log; cut -c1-260 $O/bench_per_on.json
