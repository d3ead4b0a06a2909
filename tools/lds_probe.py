"""GPU experiment: per-CG-iteration cost of the LDS-resident fused kernel (timing slope over max_CG_iter)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench

dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg2")
ITERS = int(os.environ.get('ITERS', '3'))
for Bx in (4096,):
    blk = bench.make_solver(n, cl, dl, info, dev)
    y = bench.synth_y(n, Bx, 12, 1, 0, dev)
    blk.max_ADMM_iter = 2
    blk.combined_loop(y, print_info=False)
    for K in (1, 2, 4, 8, 16, 100):
        blk.max_CG_iter = K
        blk.max_ADMM_iter = ITERS
        blk._reset_history()
        blk.prof_begin()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        blk.combined_loop(y, print_info=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = blk.prof_end()
        print(f"B={Bx} maxCG={K:3d} kernel per iteration {pr[0]['ms']/ITERS:8.3f} ms ({pr[0]['count']} launches)  wall/iter {dt/ITERS*1e3:8.3f} ms  "
              f"cg iters x/zu/zd {blk.CG_iter_x[-1].float().mean():.1f}/{blk.CG_iter_zu[-1].float().mean():.1f}/{blk.CG_iter_zd[-1].float().mean():.1f}", flush=True)
    blk.close()
