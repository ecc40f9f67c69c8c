#!/bin/bash
# sample the shader clock while 60 solves run
python3 - <<'PY' &
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass()
for _ in range(150): A.solve_batch(S, 200, want_traj=False)
PY
PID=$!
sleep 2.0
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2; sleep 0.4; done
wait $PID
echo idle:; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
