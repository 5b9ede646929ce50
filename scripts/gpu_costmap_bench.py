"""Throughput of the costmap -> point-obstacle kernel against the HBM roofline (device-resident buffers, HIP-event time of the kernel).
Algorithmic bytes per instance: size_x * size_y (costmap read once) + 20 per obstacle written."""
import json, os, sys
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_local_planner_amd as m

out = []
for (B, sx, sy, dens) in ((1024, 200, 200, 0.01), (4096, 120, 120, 0.02), (256, 1000, 1000, 0.002)):
    O = 1024
    s = m.BatchSolver(m.config_unicycle_quadratic(20, max_obstacles=O, max_vertices=1), max_batch=B)
    g = torch.Generator(device="cuda").manual_seed(1)
    cost = torch.where(torch.rand((B, sy, sx), device="cuda", generator=g) < dens, 254, 0).to(torch.uint8)
    origin = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
    pose = torch.tensor([[sx * 0.025, sy * 0.025, 0.3]], dtype=torch.float64, device="cuda").repeat(B, 1)
    no = torch.zeros(B, dtype=torch.int32, device="cuda"); nv = torch.zeros((B, O), dtype=torch.int32, device="cuda")
    vt = torch.zeros((B, O, 1, 2), dtype=torch.float64, device="cuda"); dr = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    p = lambda t: C.c_void_p(t.data_ptr())
    for it in range(3):
        rc = s._lib.mpc_costmap_to_obstacles_device(s._h, B, p(cost), sx, sy, 0.05, p(origin), p(pose), 1.5, p(no), p(nv), p(vt), p(dr))
        assert rc == 0
        s.synchronize()
    import time
    K = 50
    t0 = time.perf_counter()
    for it in range(K):
        s._lib.mpc_costmap_to_obstacles_device(s._h, B, p(cost), sx, sy, 0.05, p(origin), p(pose), 1.5, p(no), p(nv), p(vt), p(dr))
    s.synchronize()
    t = (time.perf_counter() - t0) * 1e3 / K          # back-to-back launches: per-launch time without the event/launch latency
    nobst = int(no.sum().item())
    alg = B * sx * sy + 20 * nobst
    out.append(dict(B=B, size=[sx, sy], lethal_frac=dens, obstacles=nobst, dropped=int(dr.sum().item()), kernel_ms=t, algorithmic_bytes=alg,
                    achieved_GBps=alg / t / 1e6, hbm_peak_GBps=8000.0, frac=alg / t / 1e6 / 8000.0))
    s.close()
print(json.dumps(out, indent=1))
