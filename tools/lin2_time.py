"""GPU experiment: the element-wise kernels of the streaming path vs torch's on the same vectors (cfg3 size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg3")
blk = bench.make_solver(n, cl, dl, info, dev)
y = bench.synth_y(n, B, 12, 1, 0, dev)
blk.max_ADMM_iter = 1
blk.combined_loop(y, print_info=False)
blk.prof_begin(); blk.max_ADMM_iter = 2; blk._reset_history(); blk.combined_loop(y, print_info=False); pr = blk.prof_end()
print("CG vector updates (k_rows): %.1f GB/s over %d launches" % (pr[1]["bytes"] / pr[1]["ms"] / 1e6, pr[1]["count"]))
a = torch.randn(B * 24 * n, device=dev); b = torch.randn_like(a); c = torch.empty_like(a)
def t(fn, k=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
dt = t(lambda: torch.add(a, b, out=c)); print("torch add 2r1w: %.1f GB/s" % (12 * a.numel() / dt / 1e9))
