"""Loader / builder for the C-ABI library (include/mpc_hip.h -> csrc/libmpc_hip.so).

The library is built IN-TREE with hipcc for gfx950; there is no CPU fallback: if the
shared object is missing or no HIP device is usable, callers get an exception.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

from ._abi import MpcConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libmpc_hip.so")
UBENCH_PATH = os.path.join(CSRC, "libmpc_ubench.so")     # measurement aid of bench.py (full-occupancy FMA rate), not part of the C ABI
SOURCES = ["mpc_capi.hip", "mpc_solve_inst.hip"]
HEADERS = ["mpc_solve_kernel.hpp", "mpc_core.hpp", "mpc_problem.hpp", "mpc_wave.hpp", "mpc_wave_layout.hpp", "mpc_wave_debug.hpp", "mpc_wave_rows.inc", "mpc_wave_passes.inc", "mpc_wave_sweeps.inc", "mpc_wave_pit.inc",
           "mpc_wave_step.inc", "mpc_wave_solve.inc", "mpc_dpp_blocks.inc", "mpc_costmap.hpp", "mpc_feasibility.hpp", "mpc_grid_update.hpp", os.path.join("..", "..", "include", "mpc_hip.h")]

EXPORTS = [
    "mpc_config_defaults", "mpc_create", "mpc_reset", "mpc_destroy", "mpc_solve_batch",
    "mpc_solve_batch_device", "mpc_step_batch", "mpc_step_batch_device", "mpc_set_grid_sizes", "mpc_set_via_points", "mpc_set_via_points_device", "mpc_costmap_to_obstacles", "mpc_costmap_to_obstacles_device", "mpc_last_candidates", "mpc_last_rows_dropped", "mpc_check_feasibility", "mpc_check_feasibility_device", "mpc_grid_update_device", "mpc_get_grid_sizes", "mpc_synchronize", "mpc_last_kernel_ms", "mpc_lds_bytes", "mpc_occupancy", "mpc_last_error", "mpc_version",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 library)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """hipcc --offload-arch=gfx950 -> csrc/libmpc_hip.so (cross-compiles without a GPU).

    Split build: the ABI + the small kernels (mpc_capi.hip) and one object per (arithmetic type, model) pair of the solve kernel
    (mpc_solve_inst.hip, three kernel instantiations each) are compiled in parallel, then linked.  A plain
    `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared mpc_capi.hip -o libmpc_hip.so` gives the same library from one translation unit."""
    if out is None:      # the measurement-only micro-benchmark rides along with the product library, but never stands in its way
        try:
            build_ubench(force=force, verbose=verbose)
        except (OSError, RuntimeError, subprocess.CalledProcessError) as e:
            print(f"mpc_local_planner_amd: libmpc_ubench.so not built ({e}); bench.py will report the data-sheet peaks only")
    if out is None and not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    out = out or LIB_PATH                      # another `out` (+ extra_flags): an instrumented copy next to the product library (tests)
    objdir = os.path.join(CSRC, "_obj") if out == LIB_PATH else out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    # -disable-lsr: LLVM's loop strength reduction gives every LDS pointer of the (manually software-pipelined) stage loops two or three induction variables plus a
    # re-materialised base; without it the loops keep the one pointer per stream the source has (headline kernel -3.8 %, profiles/r03_lsr_ab.log)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DMPC_SPLIT_BUILD", "-mllvm", "-disable-lsr"] + list(extra_flags)
    jobs = [(base + ["-c", os.path.join(CSRC, "mpc_capi.hip"), "-o", os.path.join(objdir, "mpc_capi.o")])]
    for t in ("double", "float"):
        for model in range(4):
            jobs.append(base + ["-DMPC_SOLVE_INST", f"-DMPC_INST_T={t}", f"-DMPC_INST_MODEL={model}", "-c", os.path.join(CSRC, "mpc_solve_inst.hip"),
                                "-o", os.path.join(objdir, f"mpc_solve_{t}_{model}.o")])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as pool:
        list(pool.map(run, jobs))
    link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + [j[-1] for j in jobs] + ["-o", out]
    run(link)
    return out


def build_ubench(force: bool = False, verbose: bool = False) -> str:
    """csrc/mpc_ubench.hip -> csrc/libmpc_ubench.so: the FMA-rate micro-benchmark bench.py reports next to the data-sheet peaks."""
    src = os.path.join(CSRC, "mpc_ubench.hip")
    if force or not os.path.exists(UBENCH_PATH) or os.path.getmtime(UBENCH_PATH) < os.path.getmtime(src):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", UBENCH_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return UBENCH_PATH


def measured_fma_peak(device: int = 0, fp64: bool = True, iters: int = 20000):
    """(TFLOP/s, ms) of the full-occupancy v_fma_f64 / v_pk_fma_f32 loop of csrc/mpc_ubench.hip on `device`; None without the library."""
    if not os.path.exists(UBENCH_PATH):
        return None
    lib = C.CDLL(UBENCH_PATH)
    fn = lib.mpc_ubench_fma_f64 if fp64 else lib.mpc_ubench_fma_f32
    fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    tf, ms = C.c_double(0.0), C.c_double(0.0)
    if fn(device, iters, C.byref(tf), C.byref(ms)) != 0:
        return None
    return tf.value, ms.value


_lib = None


def load() -> C.CDLL:
    """dlopen the C-ABI library and declare the prototypes of include/mpc_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MPC_HIP_LIB", LIB_PATH)      # developer override (e.g. an instrumented build); default: the in-tree library
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    dp = C.c_void_p   # raw addresses: host numpy buffers or device pointers
    lib.mpc_config_defaults.argtypes = [C.POINTER(MpcConfig)]
    lib.mpc_config_defaults.restype = None
    lib.mpc_create.argtypes = [C.POINTER(MpcConfig), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.mpc_create.restype = C.c_int
    lib.mpc_reset.argtypes = [C.c_void_p]
    lib.mpc_reset.restype = C.c_int
    lib.mpc_destroy.argtypes = [C.c_void_p]
    lib.mpc_destroy.restype = None
    sig = [C.c_void_p, C.c_int32] + [dp] * 7 + [C.c_void_p] + [dp] * 5     # ..., const mpc_obstacles*, outputs
    lib.mpc_solve_batch.argtypes = sig
    lib.mpc_solve_batch.restype = C.c_int
    lib.mpc_solve_batch_device.argtypes = sig
    lib.mpc_solve_batch_device.restype = C.c_int
    stp = [C.c_void_p, C.c_int32] + [dp] * 7 + [C.c_void_p] + [C.c_int32] * 4 + [C.c_double] + [dp] * 5
    lib.mpc_step_batch.argtypes = stp + [dp]
    lib.mpc_step_batch.restype = C.c_int
    lib.mpc_step_batch_device.argtypes = stp
    lib.mpc_step_batch_device.restype = C.c_int
    lib.mpc_set_grid_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.mpc_set_grid_sizes.restype = C.c_int
    lib.mpc_set_via_points.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mpc_set_via_points.restype = C.c_int
    lib.mpc_set_via_points_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mpc_set_via_points_device.restype = C.c_int
    cm = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mpc_costmap_to_obstacles.argtypes = cm
    lib.mpc_costmap_to_obstacles.restype = C.c_int
    lib.mpc_costmap_to_obstacles_device.argtypes = cm
    lib.mpc_costmap_to_obstacles_device.restype = C.c_int
    lib.mpc_last_candidates.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.mpc_last_candidates.restype = C.c_int
    lib.mpc_last_rows_dropped.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.mpc_last_rows_dropped.restype = C.c_int
    fs = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p]
    lib.mpc_check_feasibility.argtypes = fs
    lib.mpc_check_feasibility.restype = C.c_int
    lib.mpc_check_feasibility_device.argtypes = fs
    lib.mpc_check_feasibility_device.restype = C.c_int
    lib.mpc_grid_update_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double]
    lib.mpc_grid_update_device.restype = C.c_int
    lib.mpc_get_grid_sizes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.mpc_get_grid_sizes.restype = C.c_int
    lib.mpc_synchronize.argtypes = [C.c_void_p]
    lib.mpc_synchronize.restype = C.c_int
    lib.mpc_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.mpc_last_kernel_ms.restype = C.c_int
    lib.mpc_lds_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.mpc_lds_bytes.restype = C.c_int
    lib.mpc_occupancy.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.mpc_occupancy.restype = C.c_int
    lib.mpc_last_error.argtypes = []
    lib.mpc_last_error.restype = C.c_char_p
    lib.mpc_version.argtypes = []
    lib.mpc_version.restype = C.c_int32
    _lib = lib
    return lib
