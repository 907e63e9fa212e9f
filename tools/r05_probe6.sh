#!/bin/bash
O=gpurun_out/r05_probe6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MORL_AC_LN_CHAIN=1 MORL_PROBE_LN_NOPOST=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gpi_nopost -- python $R/bench_ac.py --workload gpi --steps 60 --no-cpu-baseline > /dev/null 2>&1
MORL_AC_LN_CHAIN=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_gpi_fwd -- python $R/bench_ac.py --workload gpi --steps 60 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_capql -- python $R/bench_ac.py --workload capql --steps 60 --no-cpu-baseline > /dev/null 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for d in nopost fwd; do for f in $(find $O/prof_gpi_$d -name "*kernel_stats.csv"); do echo == $d; head -4 $f | cut -c1-150; done; done
for f in $(find $O/prof_capql -name "*kernel_stats.csv"); do echo == capql; head -8 $f | cut -c1-150; done
