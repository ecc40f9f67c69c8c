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
           "ascent_dense_records", "ascent_coast_batch", "ascent_kkt_solve", "ascent_last_kernel_ms", "ascent_default_path",
           "ascent_workspace_layout")
PATHS = {"auto": 0, "fused": 1, "split_lane": 2, "split_wide": 3, "dense": 4, "persist": 5}     # enum ascent_path

_lib = None


class AscentLibraryError(RuntimeError):
    pass


def hip_runtimes_mapped() -> list[str]:
    """Paths of the HIP runtimes (libamdhip64) mapped into this process."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                if "libamdhip64" in os.path.basename(path) and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def _bind_hip_runtime():
    """One HIP runtime per process.  libascent.so needs `libamdhip64.so.7`; PyTorch-ROCm ships its own copy of that library
    (torch/lib/libamdhip64.so, same SONAME) and loads it by file name.  Whichever is mapped first wins for libascent.so
    (the loader matches its DT_NEEDED by SONAME), but torch loaded second would map ITS copy beside the system one: two
    runtimes, and a device pointer or stream of one is meaningless to the other.  So, before libascent.so is loaded and
    unless a runtime is mapped already, map the copy torch will use (found without importing torch) -- libascent.so then binds
    to it, and a later `import torch` finds its own file loaded.  ASCENT_HIP_RUNTIME=system keeps /opt/rocm's runtime (the
    device-pointer entry points then refuse to work beside an imported torch, see require_single_hip_runtime)."""
    if hip_runtimes_mapped() or os.environ.get("ASCENT_HIP_RUNTIME", "auto") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError as e:       # fall back to the system runtime; the check below still guards the device-pointer calls
            import warnings
            warnings.warn(f"could not map {cand} ({e}); libascent.so binds to the system HIP runtime")


def require_single_hip_runtime():
    """Called by the entry points that take torch's device pointers / streams."""
    m = hip_runtimes_mapped()
    if len(m) > 1:
        raise AscentLibraryError(
            "two HIP runtimes are mapped into this process (" + ", ".join(m) + "): libascent.so was bound to one before torch "
            "loaded the other, so torch's device pointers and streams cannot be handed to it.  Import this package with "
            "ASCENT_HIP_RUNTIME unset (it then maps torch's runtime first), or `import torch` before the first solve.")


def load():
    """Load libascent.so; raises AscentLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AscentLibraryError(
            f"{LIB_PATH} not found: build it with `python -m lunar_module_ascent_trajectory_optimiser_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _bind_hip_runtime()
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
    L.ascent_workspace_layout.restype = C.c_int
    L.ascent_workspace_layout.argtypes = [C.c_int64, C.POINTER(AscentOptsC), C.POINTER(C.c_int64)]
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
