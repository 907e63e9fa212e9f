O=gpurun_out/t6; mkdir -p $O
for st in 0 64 128 256 512; do
  MORL_DW_STAGGER=$st timeout 300 python bench.py --steps 200 --warmup 60 --no-cpu-baseline > $O/b_$st.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$st.json')); print('dw stagger $st: ms/step %.4f'%d['ms_per_step'])"
done
