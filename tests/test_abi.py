"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol that
include/mpc_hip.h declares, the ctypes mirror of mpc_config has the C layout, and -- without a GPU --
mpc_create fails loudly instead of falling back to a CPU path.  No compute calls here."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mpc_hip.h")


@pytest.fixture(scope="module")
def lib():
    from mpc_local_planner_amd import _lib
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        _lib.build()
    return _lib.load()


def _declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z_0-9]+)\s*\(", txt)))


def test_header_functions_all_exported(lib):
    from mpc_local_planner_amd import _lib
    names = _declared_functions()
    assert set(names) == set(_lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_config_struct_layout_matches_c(tmp_path):
    """every field of the ctypes mirror sits where the C compiler puts it (struct mpc_config and struct mpc_obstacles)."""
    from mpc_local_planner_amd._abi import MpcConfig, MpcObstacles
    fields = [f[0] for f in MpcConfig._fields_]
    ofields = [f[0] for f in MpcObstacles._fields_]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mpc_hip.h"\nint main(){\n'
                   'printf("%zu %zu\\n", sizeof(mpc_config), sizeof(mpc_obstacles));\n' +
                   "".join('printf("%%zu\\n", offsetof(mpc_config,%s));\n' % f for f in fields) +
                   "".join('printf("%%zu\\n", offsetof(mpc_obstacles,%s));\n' % f for f in ofields) + 'return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == C.sizeof(MpcConfig) and out[1] == C.sizeof(MpcObstacles)
    offs = out[2:]
    for f, o in zip(fields, offs[:len(fields)]):
        assert getattr(MpcConfig, f).offset == o, f
    for f, o in zip(ofields, offs[len(fields):]):
        assert getattr(MpcObstacles, f).offset == o, f


def test_defaults_match_reference_in_code_defaults(lib):
    from mpc_local_planner_amd._abi import MpcConfig
    c = MpcConfig()
    lib.mpc_config_defaults(C.byref(c))
    # src/controller.cpp:274,278,236,282,391,346
    assert c.n == 20 and c.dt_ref == pytest.approx(0.3) and c.dt_free == 1
    assert list(c.xf_fixed) == [1, 1, 1] and c.max_iter == 100 and c.model == 0
    assert c.du_lb[0] < -1e29 and c.du_ub[1] > 1e29
    assert lib.mpc_version() >= 100


def test_invalid_config_rejected(lib):
    from mpc_local_planner_amd._abi import config_carlike_min_time, MPC_EINVAL
    h = C.c_void_p()
    bad = config_carlike_min_time(n=2)
    assert lib.mpc_create(C.byref(bad), 4, 0, C.byref(h)) == MPC_EINVAL
    assert b"n out of range" in lib.mpc_last_error()
    bad = config_carlike_min_time(n=20)
    bad.collocation = 7          # unknown method (0 forward, 1 midpoint, 2 Crank-Nicolson exist)
    assert lib.mpc_create(C.byref(bad), 4, 0, C.byref(h)) == MPC_EINVAL
    bad = config_carlike_min_time(n=20)
    bad.dt_free = 0
    assert lib.mpc_create(C.byref(bad), 4, 0, C.byref(h)) == MPC_EINVAL


def test_no_gpu_means_loud_failure_not_cpu_fallback(lib):
    """In the CPU-only container mpc_create must return MPC_ENODEV (the product has no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    from mpc_local_planner_amd import BatchSolver, MpcError, config_carlike_min_time
    from mpc_local_planner_amd._abi import MPC_ENODEV
    with pytest.raises(MpcError) as ei:
        BatchSolver(config_carlike_min_time(20), max_batch=4)
    assert ei.value.code == MPC_ENODEV
    assert "no CPU fallback" in str(ei.value)


def test_product_package_does_not_import_the_oracle():
    code = ("import sys; import mpc_local_planner_amd; "
            "bad=[m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; print(bad); sys.exit(1 if bad else 0)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mpc_local_planner_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt.replace("oracle/se2_nlp.py)", ""), f


def test_workloads_are_deterministic_and_in_range():
    from mpc_local_planner_amd import workloads as W
    a = W.carlike_min_time_inputs(64)
    b = W.carlike_min_time_inputs(64)
    for p, q in zip(a, b):
        np.testing.assert_array_equal(p, q)
    x0, xf, up, dtp = a
    r = np.hypot(xf[:, 0], xf[:, 1])
    assert r.min() >= 1.0 and r.max() <= 6.0
    assert up[:, 0].min() >= -0.2 and up[:, 0].max() <= 0.4 and np.abs(up[:, 1]).max() <= 0.5
    assert (dtp == 0.2).all()
    c = W.carlike_min_time_inputs(64, seed=W.SEED_CONFIG2 + 1)
    assert not np.array_equal(c[0], a[0])
