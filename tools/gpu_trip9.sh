O=gpurun_out/t9; mkdir -p $O
timeout 600 python -m pytest tests/test_distributed.py tests/test_kernels_parity.py -m gpu -x -q 2>&1 | tail -5
for m in native torch; do
MORL_COMM=$m timeout 300 python bench.py --gpus 1 --force-shard --steps 100 --warmup 10 --no-cpu-baseline > $O/shard_$m.json 2> $O/shard_$m.err
python -c "
import json; d=json.load(open('$O/shard_$m.json')); print('force-shard $m: ms/step %.4f'%d['ms_per_step'])"; tail -2 $O/shard_$m.err
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print('envelope: ms/step %.4f'%(d['ms_per_step']))"
