"""GPU timing of the individual operator kernels on cfg3 (apply ops + LHS through the ABI, HIP-event profiled)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem(os.environ.get("WL", "cfg3"))
x = torch.randn(B, 24, n, 1, device=dev)
blk = bench.make_solver(n, cl, dl, info, dev)
blk.apply_op_Lu(x)
for nm, fn, tag in (("Lu", blk.apply_op_Lu, 2), ("Ldr", blk.apply_op_Ldr, 2), ("Ldr_T", blk.apply_op_Ldr_T, 2), ("LHS_zd", blk.LHS_zd, 0), ("LHS_zu", blk.LHS_zu, 0)):
    blk.prof_begin()
    for _ in range(10):
        fn(x)
    pr = blk.prof_end()
    ms = pr[tag]["ms"] / max(1, pr[tag]["count"])
    print(f"{nm:6s} avg {ms*1e3:8.1f} us  alg {pr[tag]['bytes']/max(1,pr[tag]['count'])/ms/1e6:8.1f} GB/s  launches {pr[tag]['count']}", flush=True)
