#!/bin/bash
O=gpurun_out/r05_probe16
mkdir -p $O
R=$PWD
for i in 1 2; do for v in old new; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/probe_old/libmorl_hip.so; fi
  for w in capql mosac gpipd gpi; do
    MORL_HIP_LIB=$L timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/ac_${w}_${v}_$i.json 2>/dev/null
    python -c "
import json; d=json.load(open('$O/ac_${w}_${v}_$i.json')); print('$v $w', d['ms_per_step'])"
  done
  MORL_EXACT_F32=1 MORL_HIP_LIB=$L timeout 200 python bench.py --gpus 1 --force-shard --emulate-world 8 --shard-axis batch --no-cpu-baseline --no-ramp-record --steps 100 --warmup 20 > $O/emu8_f32_${v}_$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/emu8_f32_${v}_$i.json')); print('$v emu8 exact-f32 batch axis', d['ms_per_step'])"
done; done
timeout 1500 python -m pytest tests/test_ac_agents.py tests/test_ac_kernels_parity.py tests/test_ac_fused_adam.py tests/test_gpi_agent.py tests/test_gpi_kernels_parity.py tests/test_ln_chain.py tests/test_chain_tilings.py tests/test_distributed.py tests/test_shape_fuzz.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
