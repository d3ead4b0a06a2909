"""GPU check that two builds of the library give the same BITS on the LDS path: runs a cfg2-shaped solve (B=512, 6 ADMM
iterations, float32) and a masked DGLR solve in a child process per library (MGADMM_LIB), compares x, the exported state and
the residual history.      python tools/ab_bitwise.py <lib A> <lib B>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import numpy as np, torch
import bench, mgadmm
dev = torch.device("cuda", 0)
out = {}
n, B, cl, dl, info, desc = bench.build_problem("cfg2")
for name, abl, masked in (("pred", "None", False), ("mask_dglr", "DGLR", True), ("pred_dgtv", "DGTV", False)):
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl), device=dev,
                                compute_dtype=torch.float32, ablation=abl)
    blk.check_stop = False
    blk.max_ADMM_iter = 6
    y = bench.synth_y(n, 512, 24 if masked else 12, seed=3, offset=0, device=dev)
    m = None
    if masked:
        m = (torch.rand(y.shape, generator=torch.Generator().manual_seed(4)) > 0.3).float().to(dev)
        y = y * m
    x = blk.solve(y, mask=m)[0]
    out[name + ".x"] = x.cpu().numpy()
    for k, v in blk.state.items():
        if v is not None:
            out[name + ".state." + k] = v.cpu().numpy()
    out[name + ".pres"] = np.array(blk.p_res_list); out[name + ".dres"] = np.array(blk.d_res_list)
    out[name + ".cgx"] = torch.stack(blk.CG_iter_x).numpy()
    blk.close()
np.savez(OUT, **out)
'''

def run(lib, out):
    env = dict(os.environ, MGADMM_LIB=os.path.abspath(lib))
    code = f"ROOT = {ROOT!r}\nOUT = {out!r}\n" + CHILD
    subprocess.check_call([sys.executable, "-c", code], env=env)

if __name__ == "__main__":
    import numpy as np
    a, b = sys.argv[1], sys.argv[2]
    run(a, "/tmp/ab_a.npz"); run(b, "/tmp/ab_b.npz")
    A, Bz = np.load("/tmp/ab_a.npz"), np.load("/tmp/ab_b.npz")
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k], Bz[k], equal_nan=True)
        d = float(np.abs(A[k].astype(np.float64) - Bz[k].astype(np.float64)).max()) if not same else 0.0
        print(f"{k:32s} {'identical' if same else 'DIFFERENT max abs diff %.3e' % d}")
        bad += not same
    print("ALL IDENTICAL" if bad == 0 else f"{bad} arrays differ")
    sys.exit(1 if bad else 0)
