# Round 3, GPU trip 1: the whole -m gpu suite on the new tree, smoke, the bench line (driver shape + default), the arg-max A/B
# (broadcast vs shuffle form of envelope_td_kernel at W = 64 and at the weak-scaled W = 512), one rank of eight on both axes,
# kernel-trace stats of the step under both arg-max forms.
set -x
R=$PWD
O=gpurun_out/r3t1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
MORL_TD_SHFL=1 timeout 300 python bench.py --no-cpu-baseline --no-ramp-record > $O/bench_shfl.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --force-shard --emulate-world 8 --no-cpu-baseline > $O/bench_emu8.json 2> $O/bench_emu8.err
timeout 300 python bench.py --gpus 1 --force-shard --emulate-world 8 --weights 512 --shard-axis weights --no-cpu-baseline > $O/bench_emu8_weak_bcast.json 2>/dev/null
MORL_TD_SHFL=1 timeout 300 python bench.py --gpus 1 --force-shard --emulate-world 8 --weights 512 --shard-axis weights --no-cpu-baseline > $O/bench_emu8_weak_shfl.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
MORL_TD_SHFL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env_shfl -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_emu8w_bcast -- python $R/bench.py --force-shard --emulate-world 8 --weights 512 --shard-axis weights --steps 60 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
MORL_TD_SHFL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_emu8w_shfl -- python $R/bench.py --force-shard --emulate-world 8 --weights 512 --shard-axis weights --steps 60 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $R
find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
mkdir -p $O/parity && cp -r gpurun_out/parity_observed $O/parity/ 2>/dev/null; cp gpurun_out/near_tie_flips_*.json $O/parity/ 2>/dev/null
tail -5 $O/gpu_tests.log; tail -2 $O/smoke.log; cut -c1-600 $O/bench_driver_like.json
