#!/bin/bash
# Developer tool (GPU box, through gpurun): SQ counters of the persistent kernel's launches for one workload of scripts/prof_solve.py
# (two --pmc passes; counters only, no API tracing).   gpurun -- 'bash scripts/sq_counters.sh config3'
# then here: python scripts/sq_summary.py gpurun_out/sq_config3_a gpurun_out/sq_config3_b
set -e
R=$GRAFT_REPO_ROOT
W=${1:-config3}
cd /tmp && export TMPDIR=/tmp
export PROF_WORKLOAD=$W PROF_N=2
rm -rf $R/gpurun_out/sq_${W}_a $R/gpurun_out/sq_${W}_b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/sq_${W}_a -o a -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/sq_${W}_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/sq_${W}_b -o b -- python3 $R/scripts/prof_solve.py > $R/gpurun_out/sq_${W}_b.log 2>&1
tail -1 $R/gpurun_out/sq_${W}_a.log
