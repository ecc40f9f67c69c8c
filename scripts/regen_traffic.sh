#!/bin/bash
# Developer tool (run on the GPU box through gpurun): the three rocprofv3 passes behind profiles/traffic.json.
#   gpurun -- 'bash scripts/regen_traffic.sh' ; then here: python scripts/traffic_from_pmc.py gpurun_out/tr_fetch gpurun_out/tr_write gpurun_out/tr_stats 3 batch4096
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr_stats $R/gpurun_out/tr_fetch $R/gpurun_out/tr_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_stats -o s -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/tr1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_fetch -o f -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/tr2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tr_write -o w -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/tr3.log 2>&1
grep "^path" $R/gpurun_out/tr1.log
