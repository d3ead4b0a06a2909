"""Element rate of the solver across graph sizes (round-3 verdict item 5: no cliff between the LDS-resident path and the
streaming path).  PEMS-like graphs (path + 11 % random chords, kNN k = 4 by shortest-path distance, built on the GPU), the
cfg2 signal generator, a fixed number of ADMM iterations; B = 4096 or the largest power of two whose workspace fits 60 GB.

    [MGADMM_SWEEP_SIZES=883,1000] python tools/size_sweep.py [iters] > profiles/r03/size_sweep.txt
Prints N, path, time-group width, B, ms per ADMM iteration, sample-iterations/s and ELEMENT-iterations/s (B T N / time).
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench
import mgadmm
from mgadmm import _lib, gpu_graph

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda", 0)
print(f"# {iters} ADMM iterations per solve (second solve timed), fp32, kNN-directed, ablation None; element rate = B * 24 * N * iterations / s")
print(f"{'N':>6s} {'path':>7s} {'TPG':>4s} {'B':>6s} {'ms/iter':>9s} {'sample-it/s':>12s} {'Gelem-it/s':>11s} {'CG x/zu/zd':>16s}")
prev = None
sizes = [int(v) for v in os.environ.get("MGADMM_SWEEP_SIZES", "170,256,307,341,358,420,512,600,700,883,1000,2000,5000").split(",")]
for n in sizes:
    ue, ud = bench.pems_like_graph(n, int(round(n * 1.11)), seed=0)
    cl, dl = gpu_graph.k_nearest_neighbors(n, ue, ud, 4, device=dev)
    cl = cl.to(torch.int64)
    r = math.sqrt(n / 24)
    info = dict(rho=2 * r, rho_u=3 * r, rho_d=2 * r, mu_u=1, mu_d1=2, mu_d2=1)
    B = int(os.environ.get("MGADMM_SWEEP_B", "4096"))
    while B * 24 * n * 4 * 22 > 60e9:
        B //= 2
    blk = bench.make_solver(n, cl, dl, info, dev)
    if os.environ.get("MGADMM_SWEEP_REORDER"):          # experiments: 'cluster' / 'rcm' / '0' instead of the host's choice
        v = os.environ["MGADMM_SWEEP_REORDER"]
        blk.reorder = int(v) if v.isdigit() else v
    y = bench.synth_y(n, B, 12, 1, 0, dev)
    blk.max_ADMM_iter = iters
    blk.combined_loop(y, print_info=False)
    torch.cuda.synchronize()
    blk._reset_history()
    t0 = time.perf_counter()
    blk.combined_loop(y, print_info=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = blk._solvers[(1, torch.float32)][0]
    path = "lds" if _lib.lib.mgadmm_solver_path(h, B) == _lib.PATH_LDS else "stream"
    tpg = _lib.query(h, _lib.Q_LDS_TPG) if path == "lds" else 0
    cg = "/".join(f"{float(torch.as_tensor(getattr(blk, k)[-1]).float().mean()):.1f}" for k in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd"))
    rate = B * 24 * n * iters / dt
    print(f"{n:6d} {path:>7s} {tpg:4d} {B:6d} {dt / iters * 1e3:9.3f} {B * iters / dt:12.0f} {rate / 1e9:11.3f} {cg:>16s}"
          + (f"   x{rate / prev:.2f} vs previous size" if prev else ""), flush=True)
    prev = rate
    blk.close()
    del blk, y
    torch.cuda.empty_cache()
