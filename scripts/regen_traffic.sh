#!/bin/bash
# Developer tool (run on the GPU box through gpurun): the three rocprofv3 passes per workload behind profiles/traffic.json.
#   gpurun -- 'bash scripts/regen_traffic.sh [workload ...]'   (default: every workload of scripts/prof_solve.py)
# then here, per workload W:  python scripts/traffic_from_pmc.py gpurun_out/tr_W_fetch gpurun_out/tr_W_write gpurun_out/tr_W_stats 3 W
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
WL=${@:-config3 config3_nodcost config4 hs4096 config5 config5_free config5_one}
for W in $WL; do
  export PROF_WORKLOAD=$W
  rm -rf $R/gpurun_out/tr_${W}_stats $R/gpurun_out/tr_${W}_fetch $R/gpurun_out/tr_${W}_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_${W}_stats -o s -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/tr_${W}_1.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_${W}_fetch -o f -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/tr_${W}_2.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_${W}_write -o w -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/tr_${W}_3.log 2>&1
  grep "^workload" $R/gpurun_out/tr_${W}_1.log
done
