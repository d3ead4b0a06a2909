"""cfg2-sized problem on the PHYSICAL adjacency (use_kNN=False: ragged W_u / W_d rows) instead of the kNN tables: which
k_admm_lds instance the plan picks and what it costs per ADMM iteration next to the kNN graph.   python tools/physical_time.py"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench, mgadmm
from mgadmm import _lib, utils as mu

dev = torch.device("cuda", 0)
n, B = int(os.environ.get("MGADMM_PHYS_N", "307")), 4096
ue, ud = bench.pems_like_graph(n, int(round(n * 1.11)), seed=0)
r = math.sqrt(n / 24)
info = dict(rho=2 * r, rho_u=3 * r, rho_d=2 * r, mu_u=1, mu_d1=2, mu_d2=1)
y = bench.synth_y(n, B, 12, 1, 0, dev)
for name, kw in (("kNN k=4", dict(use_kNN=True, k=4, tables=mu.k_nearest_neighbors(n, ue, ud, 4))),
                 ("physical", dict(use_kNN=False, tables=mu.connect_list(n, ue, ud)))):
    cl, dl = kw.pop("tables")
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, u_sigma=50, d_sigma=50, tables=(cl.to(torch.int64), dl), device=dev,
                                compute_dtype=torch.float32, record_cg_coeffs=False, **kw)
    blk.check_stop = False
    blk.max_ADMM_iter = 7
    blk.combined_loop(y, print_info=False)
    torch.cuda.synchronize()
    blk._reset_history()
    t0 = time.perf_counter()
    blk.combined_loop(y, print_info=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = blk._solvers[(1, torch.float32)][0]
    path = "lds" if _lib.lib.mgadmm_solver_path(h, B) == _lib.PATH_LDS else "stream"
    cg = [float(torch.stack([v.float().mean() for v in getattr(blk, k2)]).mean()) for k2 in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd")]
    print(f"{name:10s} path {path} TPG {_lib.query(h, _lib.Q_LDS_TPG)} uniform {_lib.query(h, _lib.Q_LDS_UNIFORM)} max row {cl.shape[1]}  "
          f"{dt / 7 * 1e3:7.3f} ms/iter  {B * 7 / dt:10.0f} sample-it/s  CG {cg[0]:.1f}/{cg[1]:.1f}/{cg[2]:.1f}", flush=True)
    blk.close()
