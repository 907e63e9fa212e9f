set -x
O=gpurun_out/host2
mkdir -p $O
python -m pytest tests/test_distributed.py tests/test_host_api.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3 > $O/test.log
python tools/host_profile.py --weights 64 > $O/host_w64.txt 2>&1
python tools/host_profile.py --weights 8 > $O/host_w8.txt 2>&1
python tools/host_profile.py --weights 64 --emulate-world 8 > $O/host_emu8.txt 2>&1
python bench.py --force-shard --emulate-world 8 --no-cpu-baseline > $O/bench_emu8.json 2>$O/bench_emu8.err
python bench.py --force-shard --emulate-world 2 --no-cpu-baseline > $O/bench_emu2.json 2>/dev/null
python bench.py --force-shard --emulate-world 4 --no-cpu-baseline > $O/bench_emu4.json 2>/dev/null
python bench.py --force-shard --no-cpu-baseline > $O/bench_shard1.json 2>/dev/null
MORL_COMM=torch python bench.py --force-shard --no-cpu-baseline > $O/bench_shard1_torch.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_emu8 -- python $GRAFT_REPO_ROOT/bench.py --force-shard --emulate-world 8 --steps 80 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat $O/test.log
head -30 $O/host_emu8.txt; head -3 $O/host_w64.txt $O/host_w8.txt
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['gpu_ms_per_step_events'], d['host_enqueue_ms_per_step'])"; done
