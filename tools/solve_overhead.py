"""GPU experiment: fixed per-solve overhead of ADMM_algorithm.combined_loop (cfg2, LDS path) -- intercept of wall time over K."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench

dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg2")
blk = bench.make_solver(n, cl, dl, info, dev)
y = bench.synth_y(n, B, 12, 1, 0, dev)
blk.max_ADMM_iter = 2
blk.combined_loop(y, print_info=False)
for K in (1, 2, 5, 10, 20, 20):
    blk.max_ADMM_iter = K
    blk._reset_history()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    blk.combined_loop(y, print_info=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"K={K:3d} wall {dt*1e3:8.2f} ms  per-iter {dt/K*1e3:7.3f}", flush=True)
blk._reset_history()
pr = cProfile.Profile(); pr.enable()
blk.combined_loop(y, print_info=False); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
