import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import numpy as np, torch
import bench
from mgadmm import _lib
dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg3")
B = int(os.environ.get("PB", "512"))
y = bench.synth_y(n, B, 12, 1, 0, dev)
for tile in ("0", "1"):
    os.environ["MGADMM_TILE"] = tile
    blk = bench.make_solver(n, cl, dl, info, dev)
    blk.max_ADMM_iter = 1
    yd = y.reshape(B, 12, n).contiguous()
    h, p = blk._solver(1, torch.float32, B)
    x = torch.empty(B, 24, n, device=dev)
    metrics = np.zeros((1, _lib.NMETRIC)); cg = np.zeros((1, 3, B), dtype=np.int32)
    hs = _lib.History(); hs.metrics = metrics.ctypes.data_as(C.POINTER(C.c_double)); hs.cg_iters = cg.ctypes.data_as(C.POINTER(C.c_int32))
    rc = _lib.lib.mgadmm_solve(h, yd.data_ptr(), None, 0, B, x.data_ptr(), None, C.byref(hs), None)
    torch.cuda.synchronize()
    print("tile", tile, "rc", rc, "cg x/zu/zd mean", cg[0].mean(1).tolist(), "min", cg[0].min(1).tolist(), "max", cg[0].max(1).tolist(),
          "x finite", bool(torch.isfinite(x).all()), "metrics", np.round(metrics[0], 2).tolist(), flush=True)
    blk.close()
