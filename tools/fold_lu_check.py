"""GPU check: the zu solve with the vector update folded into the Lu tile kernel gives BITWISE the iterates of the
separate-kernel form (MGADMM_FOLD_LU=0), on a cluster-ordered 2-D kNN graph through the streaming path."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))


def run():
    import numpy as np, torch, mgadmm, bench
    n, B = 2000, 256
    pts = np.random.default_rng(0).random((n, 2))
    cl, dl = mgadmm.utils.knn_from_points(pts, 4, scale=1000.0)
    import math
    r = math.sqrt(n / 24)
    info = dict(rho=2 * r, rho_u=3 * r, rho_d=2 * r, mu_u=1, mu_d1=2, mu_d2=1)
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl),
                                device=torch.device("cuda:0"), compute_dtype=torch.float32, path="stream", reorder="cluster")
    blk.max_ADMM_iter = 4
    blk.check_stop = False
    y = bench.synth_y(n, B, 12, 3, 0, torch.device("cuda:0"))
    x, (zu, zd), phi, hist = blk.solve(y)
    return x.cpu().numpy(), zu.cpu().numpy(), [int(torch.as_tensor(v).sum()) for v in blk.CG_iter_zu]


if __name__ == "__main__":
    import numpy as np
    if len(sys.argv) > 1:
        x, zu, it = run()
        np.savez(sys.argv[1], x=x, zu=zu, it=np.array(it))
        sys.exit(0)
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, MGADMM_FOLD_LU=flag)
        f = f"/tmp/fold_lu_{flag}.npz"
        subprocess.check_call([sys.executable, __file__, f], env=env)
        outs.append(np.load(f))
    same_x = np.array_equal(outs[0]["x"], outs[1]["x"]); same_zu = np.array_equal(outs[0]["zu"], outs[1]["zu"])
    print("CG_iter_zu sums", outs[0]["it"], outs[1]["it"], "bitwise x", same_x, "bitwise zu", same_zu)
    sys.exit(0 if (same_x and same_zu and np.array_equal(outs[0]["it"], outs[1]["it"])) else 1)
