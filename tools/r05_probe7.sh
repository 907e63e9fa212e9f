#!/bin/bash
O=gpurun_out/r05_probe7
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in hip probe1 probe3 probe7 probe15; do
  MORL_HIP_LIB=$R/morl-baselines_amd/lib/libmorl_$v.so MORL_AC_LN_CHAIN=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -- python $R/bench_ac.py --workload gpi --steps 60 --no-cpu-baseline > /dev/null 2>&1
  for f in $(find $R/$O/prof_$v -name "*kernel_stats.csv"); do echo == $v; grep "mlp_chain16_post_kernel" $f | cut -c1-120; done
done
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
