"""GPU experiment: data front end at PEMS04 size (16992 steps x 307 nodes, float32 and float64): per-node
statistics, normalisation, and a batch of 4096 sliding windows (the cfg2 batch) -- time and HBM GB/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
from mgadmm.dataset import TrafficDataset, gather_windows, series_stats


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for dt in (torch.float32, torch.float64):
    es = torch.finfo(dt).bits // 8
    s = (torch.randn(16992, 307, 1, dtype=dt, device="cuda") * 40 + 200)
    starts = torch.arange(4096, device="cuda")
    t = timed(lambda: series_stats(s))
    print(f"{dt}: stats (min,max,mean,std; 2 passes) {t*1e6:8.1f} us  {2*s.numel()*es/t/1e9:7.1f} GB/s")
    mn, mx, mean, std = series_stats(s)
    t = timed(lambda: TrafficDataset._affine(s, mean, std, None, False))
    print(f"{dt}: standardize in place           {t*1e6:8.1f} us  {2*s.numel()*es/t/1e9:7.1f} GB/s")
    t = timed(lambda: gather_windows(s, starts, 24))
    out_b = 4096 * 24 * 307 * es
    print(f"{dt}: 4096 windows x 24 steps        {t*1e6:8.1f} us  {out_b/t/1e9:7.1f} GB/s written ({out_b/1e6:.0f} MB; reads hit L2)")
