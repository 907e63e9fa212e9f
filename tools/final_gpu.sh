set -x
R=$PWD
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/gpu_tests.log
python __graft_entry__.py --smoke > gpurun_out/final/smoke.log 2>&1
python bench.py > gpurun_out/final/bench_per_on.json 2> gpurun_out/final/bench_per_on.err
python bench.py --per 0 --no-cpu-baseline > gpurun_out/final/bench_per_off.json 2>/dev/null || true
for w in capql mosac gpipd gpi ens; do python bench_ac.py --workload $w > gpurun_out/final/bench_ac_$w.json 2>/dev/null; done
python bench_ac.py --workload morld --pop 64 > gpurun_out/final/bench_ac_morld64.json 2>/dev/null
python bench_ac.py --workload morld --pop 128 --no-cpu-baseline > gpurun_out/final/bench_ac_morld128.json 2>/dev/null
for n in 1024 16384 65536; do python bench_front.py --workload pareto --n $n > gpurun_out/final/bench_front_pareto_$n.json 2>/dev/null; done
for r in 2 3 4; do python bench_front.py --workload hv --r $r > gpurun_out/final/bench_front_hv_r$r.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_env -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_pareto -- python $R/bench_front.py --workload pareto --n 16384 --no-cpu-baseline > /dev/null 2>&1
for w in mosac gpi ens morld; do rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_$w -- python $R/bench_ac.py --workload $w --steps 60 --no-cpu-baseline > /dev/null 2>&1; done
cd $R
tail -2 gpurun_out/final/gpu_tests.log; tail -2 gpurun_out/final/smoke.log; cut -c1-400 gpurun_out/final/bench_per_on.json
