"""GPU experiment: wall time of the GPU table builders (mgadmm.gpu_graph) vs the host builders (mgadmm.utils)
on road-like graphs (path + chords, degree ~2.4), k = 4.  Host kNN is timed on N <= 20k only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from mgadmm import gpu_graph as gg, utils as mu


def road(n, seed=0):
    rng = np.random.default_rng(seed)
    a = np.arange(n - 1)
    ch = rng.integers(0, n, size=(n // 5, 2))
    ch = ch[ch[:, 0] != ch[:, 1]]
    e = np.concatenate([np.stack([a, a + 1], 1), ch])
    d = rng.uniform(3.0, 2900.0, size=len(e))
    return torch.from_numpy(np.concatenate([e, e[:, ::-1]])), torch.from_numpy(np.concatenate([d, d]))


torch.cuda.init()
gg.k_nearest_neighbors(*([4] + list(road(4))), 1)        # warm-up (kernel load)
for n in (10_000, 100_000, 1_000_000):
    ue, ud = road(n)
    t0 = time.perf_counter(); cl, dl = gg.k_nearest_neighbors(n, ue, ud, 4); t1 = time.perf_counter()
    u, d, s = gg.weight_tables(cl.long(), dl); t2 = time.perf_counter()
    line = f"N={n:8d} E={len(ue):8d}  gpu kNN {1e3*(t1-t0):8.1f} ms  gpu weights {1e3*(t2-t1):7.1f} ms"
    if n <= 20_000:
        t0 = time.perf_counter(); clh, dlh = mu.k_nearest_neighbors(n, ue, ud, 4); t1 = time.perf_counter()
        uh = mu.undirected_graph_from_distance(clh.long(), dlh); dh = mu.directed_graph_from_distance(clh.long(), dlh); t2 = time.perf_counter()
        line += f"   host kNN {1e3*(t1-t0):8.1f} ms  host weights {1e3*(t2-t1):7.1f} ms  equal={bool((clh == cl).all())}"
    print(line, flush=True)
