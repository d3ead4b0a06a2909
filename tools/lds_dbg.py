import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from helpers import make_product, case_inputs
from mgadmm import _lib
meta = load_golden("g4_meta.npz")
y, _ = case_inputs(meta, "pred", np.float64)
for path in ("lds",):
    for (K, NI, REC) in ((100, 1, 1), (100, 1, 1), (100, 1, 1), (30, 1, 1)):
        blk = make_product(meta, "knn", path=path)
        blk.max_ADMM_iter = NI
        blk.max_CG_iter = K
        yd = torch.from_numpy(y).float().cuda().contiguous()
        h, p = blk._solver(1, torch.float32, 1)
        x = torch.empty(1, 24, 30, device="cuda")
        metrics = np.zeros((NI, _lib.NMETRIC)); cg = np.zeros((NI, 3, 1), dtype=np.int32)
        hs = _lib.History(); hs.metrics = metrics.ctypes.data_as(C.POINTER(C.c_double)); hs.cg_iters = cg.ctypes.data_as(C.POINTER(C.c_int32))
        st = _lib.State()
        al = np.zeros((NI, 3, K, 1)); be = np.zeros((NI, 3, K, 1))
        if REC:
            hs.cg_alpha = al.ctypes.data_as(C.POINTER(C.c_double)); hs.cg_beta = be.ctypes.data_as(C.POINTER(C.c_double))
        zu = torch.empty_like(x); st.zu = zu.data_ptr()
        rc = _lib.lib.mgadmm_solve(h, yd.data_ptr(), None, 0, 1, x.data_ptr(), C.byref(st), C.byref(hs), None)
        torch.cuda.synchronize()
        print(path, "NI", NI, "REC", REC, "rc", rc, "cg", cg.ravel().tolist(), "x", float(x.abs().max()), float(x.norm()), "zu", float(zu.norm()), "metrics", np.round(metrics[-1][:4], 3).tolist(), flush=True)
        print("   alpha_x", np.array2string(al[0,0,:24,0], precision=5, max_line_width=250))
        print("   beta_x ", np.array2string(be[0,0,:24,0], precision=3, max_line_width=250))
        blk.close()
