"""GPU experiment: wall time of BASELINE cfg1 (N=170, B=1, 50 iterations) on the three kernel paths, and cfg1-size batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench, mgadmm
n, B, cl, dl, info, _ = bench.build_problem("cfg1")
for Bx in (1, 64):
    y = 300 * torch.rand(Bx, 12, n, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).cuda()
    for name, kw in (("f64 stream", dict(compute_dtype=torch.float64)), ("f32 lds", dict(path="lds")), ("f32 stream", dict(path="stream"))):
        blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl), **kw)
        blk.max_ADMM_iter = 50; blk.check_stop = False
        blk.combined_loop(y, print_info=False); blk._reset_history()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        blk.combined_loop(y, print_info=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={Bx:3d} {name:10s} 50 iterations {dt*1e3:8.1f} ms  = {50*Bx/dt:9.1f} sample-it/s", flush=True)
        blk.close()
