O=gpurun_out/nmaj
mkdir -p $O
python -m pytest tests/test_ac_kernels_parity.py tests/test_ac_agents.py tests/test_chain_tilings.py tests/test_shape_fuzz.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
python bench_ac.py --workload morld --pop 64 --no-cpu-baseline > $O/m64_$rep.json 2>/dev/null
MORL_AC_NMAJOR=0 python bench_ac.py --workload morld --pop 64 --no-cpu-baseline > $O/m64_k_$rep.json 2>/dev/null
done
python bench_ac.py --workload morld --pop 128 --no-cpu-baseline > $O/m128.json 2>/dev/null
python bench_ac.py --workload morld --pop 16 --no-cpu-baseline > $O/m16.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench_ac.py --workload morld --pop 64 --steps 40 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for f in m64_1 m64_2 m64_k_1 m64_k_2 m128 m16; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['achieved'])"; done
