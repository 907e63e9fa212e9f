set -x
O=gpurun_out/t5
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for ch in 0 1; do
for w in capql mosac gpipd gpi; do
  MORL_AC_CHAIN=$ch timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/ac_${w}_chain$ch.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/ac_${w}_chain$ch.json')); print('chain=$ch $w: ms/update %.4f'%d['ms_per_step'])"
done
done
MORL_AC_CHAIN=1 timeout 300 python bench_ac.py --workload morld --pop 64 --no-cpu-baseline > $O/ac_morld64.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/ac_morld64.json')); print('morld64: ms/update %.4f'%d['ms_per_step'])"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print('envelope: ms/step %.4f host %.4f'%(d['ms_per_step'], d['host_enqueue_ms_per_step']))"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench20.json')); print('envelope 20-step: ms/step %.4f'%(d['ms_per_step']))"
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_capql -- python $R/bench_ac.py --workload capql --steps 60 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/$O/prof_capql -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-120
