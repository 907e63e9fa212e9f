#!/bin/bash
# One parametrised GPU collection script (replaces the per-trip scripts of rounds 1-3).  Run from the repo root on the GPU box:
#   tools/gpu_collect.sh <out-tag> <stage> [<stage> ...]
# stages: tests-quick | tests-all | smoke | bench | bench-exact | bench-eager | bench-driver | bench-w32 | prof-env | prof-env-exact | prof-w32 |
#         bench-ac | prof-ac | emu8 | emu-all | host-prof | morld-ctx | lazy-sweep | pmc | fronts | rank-ab | probes | target-ab
# Everything lands under gpurun_out/<out-tag>/ (merged back by gpurun); large traces are deleted, the stats CSVs kept.
set -x
R=$PWD
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
B="--no-cpu-baseline --no-ramp-record"
S="--no-sustained-record --no-exact-record"     # (the single-figure stages; bench-driver keeps every record)
prof() {   # prof <name> <command...>: rocprofv3 kernel stats of the command (run from /tmp as the guide prescribes)
    local name=$1; shift
    (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -- "$@" > /dev/null 2>&1)
    python tools/trace_phases.py $(find $O/prof_$name -name "*kernel_trace.csv" | head -1) ${OPENER:-step_prologue} > $O/step_timeline_$name.txt 2>&1 || true
    find $O/prof_$name -name "*kernel_trace.csv" -delete; find $O/prof_$name -name "*agent_info.csv" -delete
}
for st in "$@"; do
case $st in
tests-quick) timeout 1500 python -m pytest tests/test_flagship_golden.py tests/test_kernels_parity.py -m gpu -q -x -p no:cacheprovider -s 2>&1 | tail -60 > $O/gpu_tests_quick.log; tail -5 $O/gpu_tests_quick.log ;;
tests-all)   timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -80 > $O/gpu_tests.log; tail -5 $O/gpu_tests.log ;;
smoke)       timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log ;;
bench)       timeout 300 python bench.py --steps 200 --warmup 20 $B $S > $O/bench_200.json 2> $O/bench_200.err; cut -c1-400 $O/bench_200.json ;;
bench-exact) MORL_EXACT_F32=1 timeout 300 python bench.py --steps 200 --warmup 20 $B $S > $O/bench_exact_f32_200.json 2>/dev/null; cut -c1-400 $O/bench_exact_f32_200.json ;;
bench-driver) timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; cut -c1-400 $O/bench_driver_like.json ;;
bench-w32)   timeout 300 python bench.py --weights 32 --steps 200 --warmup 20 $B $S > $O/bench_w32.json 2>/dev/null; cut -c1-300 $O/bench_w32.json
             MORL_EXACT_F32=1 timeout 300 python bench.py --weights 32 --steps 200 --warmup 20 $B $S > $O/bench_w32_exact_f32.json 2>/dev/null; cut -c1-300 $O/bench_w32_exact_f32.json ;;
prof-env)    prof env python $R/bench.py --steps 80 --warmup 10 $B $S ;;
prof-env-exact) MORL_EXACT_F32=1 prof env_exact_f32 python $R/bench.py --steps 80 --warmup 10 $B $S ;;
prof-w32)    prof w32 python $R/bench.py --weights 32 --steps 80 --warmup 10 $B $S ;;
bench-ac)    for w in capql mosac gpipd gpi ens; do timeout 300 python bench_ac.py --workload $w > $O/bench_ac_$w.json 2>/dev/null; cut -c1-200 $O/bench_ac_$w.json; done
             timeout 300 python bench_ac.py --workload morld --pop 64 > $O/bench_ac_morld64.json 2>/dev/null ;;
prof-ac)     OPENER=ac_inputs prof capql python $R/bench_ac.py --workload capql --steps 60 --no-cpu-baseline
             OPENER=ac_transpose_scatter prof gpi python $R/bench_ac.py --workload gpi --steps 60 --no-cpu-baseline ;;
emu8)        timeout 300 python bench.py --gpus 1 --force-shard --emulate-world 8 $B $S > $O/bench_emulated_rank_of_8.json 2>/dev/null; cut -c1-300 $O/bench_emulated_rank_of_8.json ;;
emu-all)     # one rank of a 2 / 4 / 8-rank strong-scaled job run alone, both axes -> the strong-scaling ceiling (tools/emulated_ceiling.py)
             timeout 300 python bench.py --steps 200 --warmup 20 $B > $O/single.json $S 2>/dev/null
             for n in 2 4 8; do for ax in batch weights; do
               timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax --steps 100 --warmup 20 $B $S > $O/emu${n}_${ax}.json 2>/dev/null
             done; done
             python tools/emulated_ceiling.py $O/single.json $O $O/emulated_ceiling.json | cut -c1-400 ;;
bench-eager) MORL_LAZY_TARGETS=0 timeout 300 python bench.py --steps 200 --warmup 20 $B $S > $O/bench_eager_targets_200.json 2>/dev/null; cut -c1-300 $O/bench_eager_targets_200.json ;;
host-prof)   timeout 300 python tools/host_profile.py --steps 300 > $O/host_profile_single.txt 2>&1; head -8 $O/host_profile_single.txt ;;
morld-ctx)   timeout 300 python bench_ac.py --workload morld --pop 64 --devices 2 --no-cpu-baseline > $O/bench_ac_morld64_2ctx.json 2>/dev/null; cut -c1-300 $O/bench_ac_morld64_2ctx.json ;;
lazy-sweep)  timeout 300 python tools/lazy_sweep.py > $O/lazy_sweep.json 2> $O/lazy_sweep.err; tail -3 $O/lazy_sweep.err ;;
pmc)         (cd /tmp && export TMPDIR=/tmp && timeout 900 python $R/tools/pmc_summary.py $R/$O/pmc_summary.json > $R/$O/pmc_summary.txt 2>&1); tail -25 $O/pmc_summary.txt ;;
fronts)      for n in 1024 16384 65536; do timeout 300 python bench_front.py --workload pareto --n $n > $O/bench_front_pareto_$n.json 2>/dev/null; done
             for r in 2 3 4; do timeout 300 python bench_front.py --workload hv --r $r > $O/bench_front_hv_r$r.json 2>/dev/null; done ;;
rank-ab)     tools/rank_step_ab.sh $O/rank_ab > $O/rank_step_bfn_ab.txt 2>&1; cp $O/rank_ab/rank_step_bfn_ab.json $O/rank_step_bfn_ab.json; tail -12 $O/rank_step_bfn_ab.txt | cut -c1-200 ;;
target-ab)   for rep in 1 2; do timeout 300 python bench.py --steps 300 --warmup 20 $B --no-sustained-record --no-exact-record > $O/target_ab_chain4_$rep.json 2>/dev/null
               MORL_BFN_TARGETS=1 timeout 300 python bench.py --steps 300 --warmup 20 $B --no-sustained-record --no-exact-record > $O/target_ab_bfn_$rep.json 2>/dev/null; done ;;
dual-ab)     timeout 900 python -m pytest tests/test_chain_tilings.py -m gpu -q -x -p no:cacheprovider -k paired 2>&1 | tail -15 > $O/dual_test.log; tail -3 $O/dual_test.log
             for rep in 1 2 3; do for d in 0 1; do
               MORL_BF_DUAL=$d timeout 300 python bench.py --steps 300 --warmup 20 $B $S > $O/dual_ab_${d}_$rep.json 2>/dev/null
               python -c "import json,sys; j=json.loads(open('$O/dual_ab_${d}_$rep.json').read()); print('dual=$d', j['ms_per_step'], {k: round(v['avg_launch_us'], 1) for k, v in j['roofline']['per_kernel'].items()})"
             done; done ;;
roll-ab)     timeout 900 python -m pytest tests/test_chain_tilings.py -m gpu -q -x -p no:cacheprovider -k rolling 2>&1 | tail -15 > $O/roll_test.log; tail -3 $O/roll_test.log
             for rep in 1 2 3; do for d in 0 1; do
               MORL_BF_ROLL=$d timeout 300 python bench.py --steps 300 --warmup 20 $B $S > $O/roll_ab_${d}_$rep.json 2>/dev/null
               python -c "import json,sys; j=json.loads(open('$O/roll_ab_${d}_$rep.json').read()); print('roll=$d', j['ms_per_step'], {k: round(v['avg_launch_us'], 1) for k, v in j['roofline']['per_kernel'].items()})"
             done; done ;;
pw-ab)       timeout 600 python -m pytest tests/test_chain_tilings.py -m gpu -q -x -p no:cacheprovider -k producer_wave 2>&1 | tail -15 > $O/pw_test.log; tail -3 $O/pw_test.log
             for rep in 1 2 3; do for d in 0 1; do
               MORL_BF_PW=$((14 + d)) timeout 300 python bench.py --steps 300 --warmup 20 $B $S > $O/pw_ab_${d}_$rep.json 2>/dev/null
               python -c "import json,sys; j=json.loads(open('$O/pw_ab_${d}_$rep.json').read()); print('pw=$d', j['ms_per_step'], {k: round(v['avg_launch_us'], 1) for k, v in j['roofline']['per_kernel'].items()})"
             done; done ;;
probes)      for p in planes_probe cmp64_probe valu_mfma_probe; do [ -x tools/probes/$p ] && timeout 200 tools/probes/$p > $O/$p.txt 2>&1; done; tail -4 $O/cmp64_probe.txt ;;
*) echo "unknown stage $st" ;;
esac
done
for f in $O/prof_*/*/*kernel_stats.csv $O/prof_*/*kernel_stats.csv; do [ -f "$f" ] && { echo "== $f"; head -14 "$f" | cut -c1-160; }; done
