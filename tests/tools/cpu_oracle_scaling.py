"""Throughput of the C oracle (the CPU baseline of bench.py) versus the number of OpenMP threads on this host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import c_oracle as CO, se2_nlp as R
import mpc_local_planner_amd.workloads as W
CO.build()
oc = CO.from_nlp_config(R.config_carlike_min_time(50))
x0, xf, up, dtp = W.carlike_min_time_inputs(8192)
CO.solve_batch(oc, x0[:256], xf[:256], up[:256], dtp[:256])
print("logical cpus", os.cpu_count())
for nt in (8, 16, 32, 64, 96, 128, 192, 256):
    if nt > os.cpu_count():
        break
    t = time.perf_counter(); CO.solve_batch(oc, x0, xf, up, dtp, nthreads=nt); dt = time.perf_counter() - t
    print(f"threads {nt:4d}: {8192 / dt:9.0f} solves/s  ({8192 / dt / nt:6.1f} per thread)")
