"""Device-resident throughput versus the batch size on one GPU (kernel time by HIP events on the solver's stream, best of 2 after a warm-up launch).
    python scripts/gpu_batch_sweep.py [headline100|headline60|single|config5_fp64|config3] ...
headline100: car-like min-time n = 50, the headline's candidates with the parity-preserving caps (100/45/40/35); headline60: candidate 0 capped at 60 (r02-r04's sweep);
single: the reference path alone; config5_fp64: bicycle n = 120 with the config-5 leg's candidates; config3: unicycle n = 80 with 16 polygons, binding placement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import mpc_local_planner_amd as m
dev = torch.device("cuda", 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for what in (sys.argv[1:] or ["headline100"]):
    print(f"# {what}")
    for B in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        obstacles = None
        if what == "config5_fp64":
            if B > 16384: continue
            n = 120; inp = m.workloads.bicycle_min_time_inputs(B)
            cfg = m.config_bicycle_min_time(n, candidates=(0, 1, 2, 5), candidate_max_iter=(60, 50, 45, 40), candidate_param=(0.0, 0.0, 0.0, 2.0))
        elif what == "config3":
            if B > 16384: continue
            n = 80; x0, xf, up, dtp, obstacles = m.workloads.unicycle_obstacle_inputs(B, n_obst=16, max_vertices=6, lateral=(0.15, 0.8)); inp = (x0, xf, up, dtp)
            cfg = m.config_unicycle_quadratic(n, max_obstacles=16, max_vertices=6, max_obstacle_rows=4, max_iter=60)
        else:
            n = 50; inp = m.workloads.carlike_min_time_inputs(B)
            cand = {"headline100": dict(candidates=(0, 5, 5, 7), candidate_max_iter=(100, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)),
                    "headline60": dict(candidates=(0, 5, 5, 7), candidate_max_iter=(60, 45, 40, 35), candidate_param=(0.0, 2.0, 3.0, 1.5)), "single": {}}[what]
            cfg = m.config_carlike_min_time(n, **cand)
        d_in = [T(a) for a in inp]
        d_ob = None if obstacles is None else [T(a) for a in obstacles]
        obp = None if obstacles is None else tuple(a.data_ptr() for a in d_ob)
        xo = torch.empty((B, n, 3), dtype=torch.float64, device=dev); uo = torch.empty((B, n, 2), dtype=torch.float64, device=dev)
        do = torch.empty(B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        s = m.BatchSolver(cfg, max_batch=B)
        ms = []
        for rep in range(3):
            s.solve_device(B, d_in[0].data_ptr(), d_in[1].data_ptr(), d_in[2].data_ptr(), d_in[3].data_ptr(), None, None, None, xo.data_ptr(), uo.data_ptr(), do.data_ptr(), st.data_ptr(), it.data_ptr(), obstacles=obp)
            s.synchronize(); ms.append(s.last_kernel_ms())
        k = min(ms[1:])
        cv = float((st == 0).float().mean())
        print(f"B={B:6d}: kernel {k:8.2f} ms  {B * cv / k * 1e3:9.0f} converged solves/s   converged {cv:.4f}  iterations of all candidates per instance {s.last_candidates(B)[1].mean():.1f}  LDS {s.lds_bytes()} B", flush=True)
        s.close()
