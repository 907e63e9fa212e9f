# Round-3 GPU collection: tests, smoke, the bench lines, kernel-stats profiles and the PMC summary.
set -x
R=$PWD
O=gpurun_out/final_r3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -80 > $O/gpu_tests.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 300 python bench.py > $O/bench_per_on.json 2> $O/bench_per_on.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2>/dev/null
timeout 300 python bench.py --per 0 --no-cpu-baseline --no-ramp-record > $O/bench_per_off.json 2>/dev/null || true
MORL_LAZY_TARGETS=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_eager_targets_200.json 2>/dev/null || true
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-ramp-record > $O/bench_lazy_targets_200.json 2>/dev/null || true
timeout 300 python bench.py --weights 32 --no-cpu-baseline --no-ramp-record > $O/bench_w32.json 2>/dev/null || true
timeout 300 python bench.py --gpus 1 --force-shard --no-cpu-baseline > $O/bench_force_shard_1rank.json 2>/dev/null || true
MORL_COMM=ipc timeout 300 python bench.py --gpus 1 --force-shard --no-cpu-baseline > $O/bench_force_shard_1rank_ipc.json 2>/dev/null || true
for n in 2 4 8; do timeout 300 python bench.py --gpus 1 --force-shard --emulate-world $n --no-cpu-baseline > $O/bench_emulated_rank_of_$n.json 2>/dev/null || true; done
timeout 300 python bench.py --gpus 1 --force-shard --emulate-world 8 --weights 512 --shard-axis weights --no-cpu-baseline > $O/bench_emulated_weak_rank_of_8.json 2>/dev/null || true
for w in capql mosac gpipd gpi ens; do timeout 300 python bench_ac.py --workload $w > $O/bench_ac_$w.json 2>/dev/null; done
timeout 300 python bench_ac.py --workload morld --pop 64 > $O/bench_ac_morld64.json 2>/dev/null
timeout 300 python bench_ac.py --workload morld --pop 128 --no-cpu-baseline > $O/bench_ac_morld128.json 2>/dev/null
for n in 1024 16384 65536; do timeout 300 python bench_front.py --workload pareto --n $n > $O/bench_front_pareto_$n.json 2>/dev/null; done
for r in 2 3 4; do timeout 300 python bench_front.py --workload hv --r $r > $O/bench_front_hv_r$r.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-ramp-record > /dev/null 2>&1
for w in capql gpi; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$w -- python $R/bench_ac.py --workload $w --steps 60 --no-cpu-baseline > /dev/null 2>&1; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_emu8 -- python $R/bench.py --force-shard --emulate-world 8 --shard-axis batch --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
timeout 900 python $R/tools/pmc_summary.py $R/$O/pmc_summary.json > $R/$O/pmc_summary.txt 2>&1
cd $R
python tools/trace_phases.py $(find $O/prof_env -name "*kernel_trace.csv") > $O/step_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
tail -3 $O/gpu_tests.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench_per_on.json; tail -20 $O/pmc_summary.txt
