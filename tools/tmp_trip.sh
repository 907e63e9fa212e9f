O=gpurun_out/nmaj2
mkdir -p $O
python -m pytest tests/test_ac_kernels_parity.py tests/test_ac_agents.py tests/test_chain_tilings.py tests/test_shape_fuzz.py tests/test_train_traces.py -m gpu -x -q 2>&1 | tail -2
for w in capql mosac gpipd; do for nm in 2 1; do MORL_AC_NMAJOR=$nm python bench_ac.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w nmajor=$nm', d['ms_per_step'])"; done; done
python bench_ac.py --workload morld --pop 16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('morld16', d['ms_per_step'])"
python bench_ac.py --workload morld --pop 64 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('morld64', d['ms_per_step'])"
