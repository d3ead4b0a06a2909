"""Diagnostic for the chunked LDS schedule: cfg2 solve with a given batch / iteration count, reports where the history stops being finite.
    python tools/chunk_diag.py <B> <iters>      (environment: MGADMM_LDS_CHUNK, MGADMM_LDS_RAGGED, MGADMM_LDS_ASYNC ...)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

B, iters = int(sys.argv[1]), int(sys.argv[2])
n, _, cl, dl, info, _ = bench.build_problem("cfg2")
dev = torch.device("cuda", 0)
blk = bench.make_solver(n, cl, dl, info, dev)
y = bench.synth_y(n, B, 12, seed=1, offset=0, device=dev)
blk.max_ADMM_iter = iters
try:
    x = blk.solve(y, print_info=False, per_sample_history=True)[0]
    print("ok finite", bool(torch.isfinite(x).all()), "p_res[0]", blk.p_res_list[0], "p_res[-1]", blk.p_res_list[-1],
          "cg", [float(torch.as_tensor(v).float().mean()) for v in (blk.CG_iter_x[-1], blk.CG_iter_zu[-1], blk.CG_iter_zd[-1])])
except AssertionError as e:
    print("FAILED", str(e)[:80])
    for nm in ("p_res_list", "d_res_list", "x_shift_list"):
        print(nm, getattr(blk, nm)[:4])
    m = blk.metrics_per_sample        # (iters, NMETRIC, B)
    names = "XSHIFT PRI_ZU DUAL_ZU PRI_PHI DUAL_PHI PRI_ZD DUAL_ZD GLR DGTV DGLR RECOVER".split()
    for it in range(min(2, m.shape[0])):
        print("iteration", it, {nm: int((~np.isfinite(m[it, k])).sum()) for k, nm in enumerate(names)}, "non-finite samples of", m.shape[2])
        print("   sample 0:", {nm: float(m[it, k, 0]) for k, nm in enumerate(names)})
        print("   cg iters sample 0..7:", [torch.as_tensor(v[it]).flatten()[:8].tolist() for v in (blk.CG_iter_x, blk.CG_iter_zu, blk.CG_iter_zd)])
