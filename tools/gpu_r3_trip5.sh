O=gpurun_out/r3t5; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed.py tests/test_flagship_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $O/tests.log
MORL_COMM=ipc timeout 200 python bench.py --force-shard --no-cpu-baseline --steps 50 > $O/bench_ipc_1rank.json 2> $O/bench_ipc_1rank.err
tail -8 $O/tests.log; cut -c1-400 $O/bench_ipc_1rank.json; tail -3 $O/bench_ipc_1rank.err
