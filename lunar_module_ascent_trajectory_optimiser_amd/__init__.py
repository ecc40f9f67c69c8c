"""MI355X-native batched lunar-ascent trajectory optimiser.

The package holds only the hot path of Ben-Bussch/Lunar_Module_Ascent_Trajectory_Optimiser
(the NLP solve behind Launch_Optimiser.py:177): csrc/ (HIP kernels + C ABI, include/ascent.h)
and the host-side mirror of the reference's problem-definition surface.
"""
from .params import AscentParams, sweep_isp_drymass, sweep_config4, PARAM_FIELDS  # noqa: F401
from .solver import (solve_batch, solve_batch_torch, last_kernel_ms, eval_nodes, kkt_step, BatchResult,  # noqa: F401
                     TRAJ_FIELDS, blob_rows, dense_records, coast_batch, kkt_solve, default_path)

__all__ = ["AscentParams", "sweep_isp_drymass", "sweep_config4", "solve_batch", "eval_nodes", "kkt_step",
           "BatchResult", "TRAJ_FIELDS", "PARAM_FIELDS", "blob_rows"]
