#!/bin/bash
# Developer tool: a variant build of the library into dbglib/ (git-ignored, travels with gpurun):
#   scripts/build_variant.sh NAME [extra hipcc flags, e.g. -DPERSIST_PROFILE] -- only the persistent kernels are recompiled,
#   the other objects come from csrc/_obj (python -m lunar_module_ascent_trajectory_optimiser_amd.build first).
set -e
name=$1; shift
C=lunar_module_ascent_trajectory_optimiser_amd/csrc
mkdir -p dbglib/_obj_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include"
for s in ${SRCS:-ascent_persist ascent_hs}; do
  /opt/rocm/bin/hipcc $F "$@" -c $C/$s.hip -o dbglib/_obj_$name/$s.o &
done
wait
objs=""
for s in ascent_solver ascent_pipeline ascent_dense ascent_blocktri ascent_persist ascent_hs; do
  if [ -f dbglib/_obj_$name/$s.o ]; then objs="$objs dbglib/_obj_$name/$s.o"; else objs="$objs $C/_obj/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o dbglib/libascent_$name.so $objs
echo dbglib/libascent_$name.so
