"""GPU experiment: cfg2 throughput when the caller hands over HOST tensors (the reference's usage: CPU y in, CPU x out),
i.e. including the PCIe copies the Python host performs around the device-resident ABI call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg2")
blk = bench.make_solver(n, cl, dl, info, dev)
y_dev = bench.synth_y(n, B, 12, 1, 0, dev)
for label, y in (("device tensors", y_dev), ("host tensors (pageable)", y_dev.cpu()), ("host tensors (pinned)", y_dev.cpu().pin_memory())):
    for K in (50,):
        blk.max_ADMM_iter = K
        blk._reset_history(); blk.combined_loop(y, print_info=False)
        blk._reset_history()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = blk.combined_loop(y, print_info=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{label:26s} K={K}: {dt*1e3:8.1f} ms  {B*K/dt:12,.0f} sample-iterations/s  (x on {x.device})", flush=True)
