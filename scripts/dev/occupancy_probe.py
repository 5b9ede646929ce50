"""developer experiment: what a second wave per SIMD buys.  The fp32 instantiation at n = 50 has a 20 KB LDS record (8 workgroups per CU) but 285 registers (1 wave per SIMD);
built with -DMPC_WAVES_PER_EU=2 the compiler spills down to 256 and two waves fit.  Same solves, kernel time at several batch sizes; run once with the product library and once with
MPC_HIP_LIB pointing at the -DMPC_WAVES_PER_EU=2 build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mpc_local_planner_amd as m
from mpc_local_planner_amd import _abi as A
out = []
for prec, tag in ((A.FP32, "fp32"), (A.FP64, "fp64")):
    for B in (1024, 8192, 32768):
        inp = m.workloads.carlike_min_time_inputs(B)
        s = m.BatchSolver(m.config_carlike_min_time(50, precision=prec, tol=1e-4 if prec == A.FP32 else 1e-8, max_iter=60), max_batch=B)
        r = s.solve(*inp); r = s.solve(*inp)
        out.append(f"{tag} B={B}: {s.last_kernel_ms():.2f} ms conv {np.mean(r.status == 0):.3f} it {r.iters.mean():.1f}")
        s.close()
print(os.environ.get("MPC_HIP_LIB", "product library"), "|", " | ".join(out))
