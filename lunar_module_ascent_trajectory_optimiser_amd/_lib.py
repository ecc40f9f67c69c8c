"""ctypes binding of csrc/libascent.so (declared in include/ascent.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing
or a symbol is absent the import of the solver fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libascent.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class AscentParamsC(C.Structure):
    """struct ascent_params (include/ascent.h), 16 doubles."""
    _fields_ = [(n, C.c_double) for n in (
        "G", "M", "R0", "Ft", "M0", "mdot", "fuel_mass", "mass_scalar", "ang_acc_max", "r_peri",
        "r_apo", "T_scale", "angle_ub", "tf_lb", "tf_ub", "dcost")]


class AscentOptsC(C.Structure):
    """struct ascent_opts (include/ascent.h)."""
    _fields_ = [("n_nodes", C.c_int32), ("scheme", C.c_int32), ("max_iter", C.c_int32),
                ("warm_start", C.c_int32), ("tol", C.c_double), ("mu_init", C.c_double),
                ("formulation", C.c_int32), ("coarse_nodes", C.c_int32), ("terminal", C.c_int32),
                ("solver_path", C.c_int32), ("move_penalty", C.c_int32), ("reserved", C.c_int32)]


SYMBOLS = ("ascent_version", "ascent_device_count", "ascent_strerror", "ascent_solve_batch",
           "ascent_eval_nodes", "ascent_kkt_step", "ascent_eval_nodes_path", "ascent_kkt_step_path",
           "ascent_dense_records", "ascent_coast_batch", "ascent_kkt_solve", "ascent_last_kernel_ms", "ascent_default_path")
PATHS = {"auto": 0, "fused": 1, "split_lane": 2, "split_wide": 3, "dense": 4, "persist": 5}     # enum ascent_path

_lib = None


class AscentLibraryError(RuntimeError):
    pass


def load():
    """Load libascent.so; raises AscentLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AscentLibraryError(
            f"{LIB_PATH} not found: build it with `python -m lunar_module_ascent_trajectory_optimiser_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(L, s):
            raise AscentLibraryError(f"{LIB_PATH} does not export {s}")
    L.ascent_version.restype = C.c_int
    L.ascent_device_count.restype = C.c_int
    L.ascent_strerror.restype = C.c_char_p
    L.ascent_strerror.argtypes = [C.c_int]
    L.ascent_last_kernel_ms.restype = C.c_double
    L.ascent_last_kernel_ms.argtypes = [C.c_int]
    L.ascent_solve_batch.restype = C.c_int
    L.ascent_solve_batch.argtypes = [C.c_void_p, C.c_int64, C.POINTER(AscentOptsC), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.ascent_eval_nodes.restype = C.c_int
    L.ascent_eval_nodes.argtypes = [C.c_void_p, C.c_int64, C.POINTER(AscentOptsC), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int]
    L.ascent_kkt_step.restype = C.c_int
    L.ascent_kkt_step.argtypes = [C.c_void_p, C.c_int64, C.POINTER(AscentOptsC), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ascent_eval_nodes_path.restype = C.c_int
    L.ascent_eval_nodes_path.argtypes = L.ascent_eval_nodes.argtypes + [C.c_int]
    L.ascent_kkt_step_path.restype = C.c_int
    L.ascent_kkt_step_path.argtypes = L.ascent_kkt_step.argtypes + [C.c_int]
    L.ascent_default_path.restype = C.c_int
    L.ascent_default_path.argtypes = [C.c_int64, C.POINTER(AscentOptsC)]
    L.ascent_dense_records.restype = C.c_int
    L.ascent_dense_records.argtypes = [C.c_void_p, C.c_int64, C.POINTER(AscentOptsC), C.c_void_p, C.c_void_p, C.c_int]
    L.ascent_kkt_solve.restype = C.c_int
    L.ascent_kkt_solve.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7 + [C.c_int, C.c_int]
    L.ascent_coast_batch.restype = C.c_int
    L.ascent_coast_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_int]
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise AscentLibraryError(f"libascent error {rc}: {load().ascent_strerror(rc).decode()}")
