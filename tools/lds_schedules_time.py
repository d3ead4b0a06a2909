"""GPU experiment: wall time of one cfg2 solve (N=307, B=4096, 20 ADMM iterations, LDS path) under the three schedules of the
outer loop (csrc/engine.h, solve_lds): SYNC (MGADMM_LDS_ASYNC=0), DEVSTOP (check_stop, stop test on the device) and
OVERLAP (fixed iteration count, metric kernels on the helper stream).  ADMM_tol is set below reach so that all 20 run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench

dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg2")
y = bench.synth_y(n, B, 12, 1, 0, dev)
for rep in range(2):
    for env in ("0", "1"):
        for cs in (True, False):
            os.environ["MGADMM_LDS_ASYNC"] = env
            blk = bench.make_solver(n, cl, dl, info, dev)
            blk.check_stop = cs
            blk.ADMM_tol = 1e-30
            blk.max_ADMM_iter = 3
            blk.combined_loop(y, print_info=False)
            blk.max_ADMM_iter = 20
            ts = []
            for _ in range(3):
                blk._reset_history()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                blk.combined_loop(y, print_info=False)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            name = "SYNC" if env == "0" else ("DEVSTOP" if cs else "OVERLAP")
            print(f"MGADMM_LDS_ASYNC={env} check_stop={cs!s:5s} {name:8s} 20 iterations: {min(ts)*1e3:7.2f} ms = {B*20/min(ts)/1e6:.3f} M sample-iterations/s", flush=True)
            blk.close()
