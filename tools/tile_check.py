"""GPU check: LDS-tiled spatial kernel vs the plain row kernel on the 10k-node graph (same cluster order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem(os.environ.get("WL", "cfg3"))
B = int(os.environ.get("PB", "300"))
x = torch.randn(B, 24, n, 1, device=dev)
res = {}
for tile in ("0", "1"):
    os.environ["MGADMM_TILE"] = tile
    blk = bench.make_solver(n, cl, dl, info, dev)
    res[tile] = [getattr(blk, "apply_op_" + nm)(x) for nm in ("Lu", "Ldr", "Ldr_T", "cLdr")] + [blk.LHS_x(x), blk.LHS_zu(x)]
    blk.close()
for k, nm in enumerate(("Lu", "Ldr", "Ldr_T", "cLdr", "LHS_x", "LHS_zu")):
    a, b = res["0"][k], res["1"][k]
    print(nm, "finite", bool(torch.isfinite(b).all()), "rel diff", float((a - b).norm() / a.norm()), "max abs", float((a - b).abs().max()))
