"""Multi-rank path on CPU: world_size-2 (and 3) gloo process groups exercise the batch sharding and the
final gather of mgadmm.dist.sharded_solve.  The per-rank solve is the CPU oracle here (test
infrastructure; the product path runs the HIP solver on each rank's GPU) -- what is under test is the
partitioning, the absence of any collective before the gather, and the reassembly order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_oracle
    from mgadmm.dist import shard_bounds, sharded_solve
    meta = load_golden("g4_meta.npz")
    g = load_golden("g5_batched.npz")
    y = torch.from_numpy(g["y"][:B])
    calls = []

    def solve_fn(ys, ms):
        calls.append(ys.shape[0])
        o = make_oracle(meta, "knn")
        o.max_ADMM_iter = 3
        return torch.from_numpy(o.combined_loop(ys.numpy()))

    x = sharded_solve(solve_fn, y)
    lo, hi = shard_bounds(B, world, rank)
    assert calls == ([hi - lo] if hi > lo else [])
    xs = sharded_solve(solve_fn, y, gather=False)
    # sub-blocks with asynchronous gathers: same result, one solve call per non-empty sub-block
    calls.clear()
    xc = sharded_solve(solve_fn, y, chunks=3)
    assert torch.equal(xc, x)
    assert sum(calls) == hi - lo and len(calls) == min(3, hi - lo)
    assert torch.equal(sharded_solve(solve_fn, y, gather=False, chunks=2), xs) if xs is not None else True
    np.save(os.path.join(out_dir, f"x_rank{rank}.npy"), x.numpy())
    if xs is not None:
        np.save(os.path.join(out_dir, f"shard_rank{rank}.npy"), xs.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (2, 5), (3, 7)])
def test_sharded_solve_gloo(tmp_path, world, B):
    port = 29500 + (os.getpid() % 400) + world * 7 + B
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    from helpers import make_oracle
    meta = load_golden("g4_meta.npz")
    g = load_golden("g5_batched.npz")
    o = make_oracle(meta, "knn")
    o.max_ADMM_iter = 3
    ref = o.combined_loop(g["y"][:B])
    from mgadmm.dist import shard_bounds
    for r in range(world):
        x = np.load(tmp_path / f"x_rank{r}.npy")
        assert x.shape == ref.shape
        np.testing.assert_allclose(x, ref, rtol=1e-12)          # samples are independent: sharding changes nothing
        lo, hi = shard_bounds(B, world, r)
        np.testing.assert_allclose(np.load(tmp_path / f"shard_rank{r}.npy"), ref[lo:hi], rtol=1e-12)


def test_sharded_solve_without_process_group():
    from mgadmm.dist import sharded_solve
    y = torch.arange(24.0).reshape(4, 6, 1, 1)
    x = sharded_solve(lambda ys, ms: ys * 2, y)
    assert torch.equal(x, y * 2)
