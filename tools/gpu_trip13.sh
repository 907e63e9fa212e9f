O=gpurun_out/t13; mkdir -p $O
for w in 4 8 16; do
  MORL_TD_WAVES=$w timeout 300 python bench.py --steps 200 --warmup 60 --no-cpu-baseline > $O/b_$w.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$w.json')); print('td waves $w: ms/step %.4f'%d['ms_per_step'])"
done
