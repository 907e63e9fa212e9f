# Round-end GPU collection (round 2): tests, smoke, the bench lines, kernel-stats profiles and the PMC summary.
set -x
R=$PWD
O=gpurun_out/final
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/gpu_tests.log
python __graft_entry__.py --smoke > $O/smoke.log 2>&1
python bench.py > $O/bench_per_on.json 2> $O/bench_per_on.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_like.json 2>/dev/null
python bench.py --per 0 --no-cpu-baseline > $O/bench_per_off.json 2>/dev/null || true
python bench.py --gpus 1 --force-shard --no-cpu-baseline > $O/bench_force_shard_1rank.json 2>/dev/null || true
python bench.py --weights 8 --no-cpu-baseline > $O/bench_shard_sized_w8.json 2>/dev/null || true
for n in 2 4 8; do python bench.py --gpus 1 --force-shard --emulate-world $n --no-cpu-baseline > $O/bench_emulated_rank_of_$n.json 2>/dev/null || true; done
for n in 2 4 8; do python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis weights --no-cpu-baseline > $O/bench_emulated_rank_of_${n}_weight_axis.json 2>/dev/null || true; done
python bench.py --gpus 1 --force-shard --shard-axis weights --no-cpu-baseline > $O/bench_force_shard_1rank_weight_axis.json 2>/dev/null || true
python tools/host_profile.py --weights 64 --emulate-world 8 > $O/host_profile_emulated_rank_of_8.txt 2>&1 || true
for w in capql mosac gpipd gpi ens; do python bench_ac.py --workload $w > $O/bench_ac_$w.json 2>/dev/null; done
python bench_ac.py --workload morld --pop 64 > $O/bench_ac_morld64.json 2>/dev/null
python bench_ac.py --workload morld --pop 128 --no-cpu-baseline > $O/bench_ac_morld128.json 2>/dev/null
for n in 1024 16384 65536; do python bench_front.py --workload pareto --n $n > $O/bench_front_pareto_$n.json 2>/dev/null; done
for r in 2 3 4; do python bench_front.py --workload hv --r $r > $O/bench_front_hv_r$r.json 2>/dev/null; done
./tools/probes/chain2_probe > $O/chain2_probe.txt 2>&1 || true
./tools/probes/dw_probe > $O/dw_probe.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_env -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
for w in capql mosac gpi ens morld; do rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$w -- python $R/bench_ac.py --workload $w --steps 60 --no-cpu-baseline > /dev/null 2>&1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_emu8 -- python $R/bench.py --force-shard --emulate-world 8 --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_pareto -- python $R/bench_front.py --workload pareto --n 16384 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/$O/pmc_summary.json > $R/$O/pmc_summary.txt 2>&1
cd $R
tail -2 $O/gpu_tests.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench_per_on.json; cat $O/pmc_summary.txt
