#!/bin/bash
# A/B: the 16-row chain kernels pull their chain's share of the argument block into the scalar cache at entry (old = the commit before)
O=gpurun_out/r05_probe18
mkdir -p $O
R=$PWD
for v in old new old new; do
  if [ $v = new ]; then L=$R/morl-baselines_amd/lib/libmorl_hip.so; else L=$R/morl-baselines_amd/lib/probe_old/libmorl_hip.so; fi
  for w in capql mosac gpipd gpi; do
    MORL_HIP_LIB=$L timeout 300 python bench_ac.py --workload $w --no-cpu-baseline > $O/ac_${w}_$v.json 2>/dev/null
    python -c "
import json; d=json.load(open('$O/ac_${w}_$v.json')); print('$v $w', d['ms_per_step'])"
  done
done
timeout 900 python -m pytest tests/test_ac_agents.py tests/test_ac_kernels_parity.py tests/test_ac_fused_adam.py tests/test_gpi_agent.py tests/test_ln_chain.py tests/test_distributed.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
