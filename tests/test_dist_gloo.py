"""Multi-rank path on CPU: world_size-2 (and 3) gloo process groups exercise the batch sharding, the final gather
of mgadmm.dist.sharded_solve (to one rank, as BASELINE.json's north star names it, or to all) and the re-forming of the
reference-style whole-batch residual history from the per-shard histories (mgadmm.dist.gather_history).  The per-rank solve is the CPU oracle here (test
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
    from mgadmm.dist import gather_history, shard_bounds, sharded_solve
    meta = load_golden("g4_meta.npz")
    g = load_golden("g5_batched.npz")
    y = torch.from_numpy(g["y"][:B])
    calls = []
    last = {}

    def solve_fn(ys, ms):
        calls.append(ys.shape[0])
        o = make_oracle(meta, "knn")
        o.max_ADMM_iter = 3
        last["o"] = o
        return torch.from_numpy(o.combined_loop(ys.numpy()))

    x = sharded_solve(solve_fn, y, gather="all")
    assert torch.equal(sharded_solve(solve_fn, y, gather=True), x)          # True has always meant: everywhere
    calls.clear()
    sharded_solve(solve_fn, y, gather="all")
    lo, hi = shard_bounds(B, world, rank)
    assert calls == ([hi - lo] if hi > lo else [])
    # the per-shard residual history, combined on rank 0: sqrt(sum local^2) / sample-weighted means
    hist = last["o"].hist if hi > lo else make_oracle(meta, "knn").hist
    gh = gather_history(hist, hi - lo)
    assert (gh is not None) == (rank == 0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "hist.npz"), **{k: np.asarray(v) for k, v in gh.items() if k.endswith("_list")},
                 iters=np.array(gh["iters_per_shard"]), samples=np.array(gh["samples_per_shard"]))
    # a list that survived an earlier solve (recover_list is kept by init_iterations, ADMM.py:100-133) is longer than the
    # others: only its last rows belong to this solve, and every list travels with its own row count
    class _H:
        pass
    hx = _H()
    for k in ("p_res_list", "d_res_list", "x_shift_list", "GLR_list", "DGTV_list", "DGLR_list", "delta_x_per_step"):
        setattr(hx, k, getattr(hist, k))
    hx.recover_list = [123.0, 456.0] + list(hist.recover_list)
    gx = gather_history(hx, hi - lo)
    if rank == 0:
        for k in gh:
            if k.endswith("_list"):
                assert torch.equal(torch.as_tensor(gx[k]), torch.as_tensor(gh[k])), k
    # sub-blocks (sharded_solve(chunks=)): the rank's history holds one block of rows per sub-block; combined like shards
    per_call = []

    def solve_keep(ys, ms):
        o = make_oracle(meta, "knn")
        o.max_ADMM_iter = 3
        per_call.append(o)
        return torch.from_numpy(o.combined_loop(ys.numpy()))

    sharded_solve(solve_keep, y, gather=False, chunks=2)
    hc = _H()
    for k in ("p_res_list", "d_res_list", "x_shift_list", "recover_list", "GLR_list", "DGTV_list", "DGLR_list", "delta_x_per_step"):
        setattr(hc, k, [row for o in per_call for row in getattr(o.hist, k)] if per_call else getattr(hist, k))
    gc = gather_history(hc, hi - lo, chunks=2)
    if rank == 0:
        for k in gh:
            if k.endswith("_list"):
                np.testing.assert_allclose(np.asarray(gc[k]), np.asarray(gh[k]), rtol=1e-10, err_msg=k)
    # default: ONE gather to rank 0 -- only rank 0 holds the full tensor
    xr = sharded_solve(solve_fn, y)
    assert (xr is not None) == (rank == 0)
    if rank == 0:
        assert torch.equal(xr, x)
    xs = sharded_solve(solve_fn, y, gather=False)
    # sub-blocks with asynchronous gathers: same result, one solve call per non-empty sub-block
    calls.clear()
    xc = sharded_solve(solve_fn, y, gather="all", chunks=3)
    assert torch.equal(xc, x)
    assert sum(calls) == hi - lo and len(calls) == min(3, hi - lo)
    xc0 = sharded_solve(solve_fn, y, chunks=2, dst=world - 1)
    assert (xc0 is not None) == (rank == world - 1) and (xc0 is None or torch.equal(xc0, x))
    assert torch.equal(sharded_solve(solve_fn, y, gather=False, chunks=2), xs) if xs is not None else True
    np.save(os.path.join(out_dir, f"x_rank{rank}.npy"), x.numpy())
    if xs is not None:
        np.save(os.path.join(out_dir, f"shard_rank{rank}.npy"), xs.numpy())
    # early stop is per shard (no collective on the convergence path): with a loose tolerance every shard stops on
    # ITS OWN norms, i.e. it equals the solve of that block alone; the combined history covers the common iterations
    def solve_stop(ys, ms):
        o = make_oracle(meta, "knn")
        o.max_ADMM_iter, o.ADMM_tol = 40, 60.0
        last["o"] = o
        return torch.from_numpy(o.combined_loop(ys.numpy()))

    xe = sharded_solve(solve_stop, y, gather=False)
    if xe is not None:
        np.save(os.path.join(out_dir, f"stop_rank{rank}.npy"), xe.numpy())
    ge = gather_history(last["o"].hist if hi > lo else make_oracle(meta, "knn").hist, hi - lo)
    if rank == 0:
        np.savez(os.path.join(out_dir, "hist_stop.npz"), iters=np.array(ge["iters_per_shard"]), p=np.asarray(ge["p_res_list"]))
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
    # the combined history equals the history of the unsharded solve (whole-batch Frobenius norms, ADMM.py:612-636)
    h = np.load(tmp_path / "hist.npz")
    assert h["iters"].tolist() == [3] * world and int(h["samples"].sum()) == B
    np.testing.assert_allclose(h["p_res_list"], np.array(o.hist.p_res_list), rtol=1e-10)
    np.testing.assert_allclose(h["d_res_list"], np.array(o.hist.d_res_list), rtol=1e-10)
    np.testing.assert_allclose(h["x_shift_list"][:, 0], np.array(o.hist.x_shift_list), rtol=1e-10)
    np.testing.assert_allclose(h["recover_list"][:, 0], np.array(o.hist.recover_list), rtol=1e-10)
    for k in ("GLR_list", "DGTV_list", "DGLR_list"):
        np.testing.assert_allclose(h[k][:, 0], np.array(getattr(o.hist, k)), rtol=1e-10)
    # early stop enabled: every shard equals the stand-alone solve of its block; iteration counts may differ per shard
    hs = np.load(tmp_path / "hist_stop.npz")
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        oo = make_oracle(meta, "knn")
        oo.max_ADMM_iter, oo.ADMM_tol = 40, 60.0
        alone = oo.combined_loop(g["y"][lo:hi])
        np.testing.assert_allclose(np.load(tmp_path / f"stop_rank{r}.npy"), alone, rtol=1e-12)
        assert hs["iters"][r] == len(oo.hist.p_res_list)
    assert hs["p"].shape[0] == hs["iters"].min()
    if (world, B) == (2, 8):
        assert len(set(hs["iters"].tolist())) > 1 and hs["iters"].max() < 40      # shards really stop at different iterations


def test_sharded_solve_without_process_group():
    from mgadmm.dist import sharded_solve
    y = torch.arange(24.0).reshape(4, 6, 1, 1)
    x = sharded_solve(lambda ys, ms: ys * 2, y)
    assert torch.equal(x, y * 2)


class _StepwiseOracle:
    """The interface sharded_solve_global_stop needs from a rank's solver (one ADMM iteration per call, resumed from the state
    of the previous call), played by the CPU oracle: call k re-runs k iterations from the initial guess with the stop test
    off -- k calls of one iteration ARE one call of k iterations -- and reports the residuals of the last one."""

    def __init__(self, meta, tol):
        from helpers import make_oracle
        self.o = make_oracle(meta, "knn")
        self.ablation, self.max_ADMM_iter, self.check_stop, self.ADMM_tol = "None", 40, True, tol
        self.p_res_list, self.d_res_list, self.state, self.calls = [], [], None, 0

    def solve(self, ys, mask=None, return_state=True, warm_start=None):
        assert self.max_ADMM_iter == 1 and not self.check_stop
        assert (warm_start is None) == (self.calls == 0)
        self.calls += 1
        self.o.ADMM_tol = 0.0
        x = self.o.combined_loop(ys.numpy(), n_iters=self.calls)
        self.p_res_list.append(self.o.hist.p_res_list[-1])
        self.d_res_list.append(self.o.hist.d_res_list[-1])
        self.state = {"x": x}
        return torch.from_numpy(x), None, None, None


def _worker_global_stop(rank, world, port, B, tol, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mgadmm.dist import shard_bounds, sharded_solve_global_stop
    meta = load_golden("g4_meta.npz")
    y = torch.from_numpy(load_golden("g5_batched.npz")["y"][:B])
    blk = _StepwiseOracle(meta, tol)
    x, n_it = sharded_solve_global_stop(blk, y, gather="all")
    lo, hi = shard_bounds(B, world, rank)
    assert blk.calls == (n_it if hi > lo else 0)
    assert (blk.max_ADMM_iter, blk.check_stop) == (40, True)          # restored
    np.save(os.path.join(out_dir, f"gx_rank{rank}.npy"), x.numpy())
    np.save(os.path.join(out_dir, f"gn_rank{rank}.npy"), np.array([n_it]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (3, 7)])
def test_sharded_solve_global_stop_gloo(tmp_path, world, B):
    """The reference's stop test on whole-batch norms across shards (mgadmm.dist.sharded_solve_global_stop): every rank stops
    after the iteration the UNSHARDED run stops after and the gathered x is the unsharded x -- unlike the per-shard stop of
    sharded_solve, where the shards of the same problem stop at different iterations (test above)."""
    tol = 60.0
    port = 29950 + (os.getpid() % 300) + world * 5 + B
    mp.spawn(_worker_global_stop, args=(world, port, B, tol, str(tmp_path)), nprocs=world, join=True)
    from helpers import make_oracle
    meta = load_golden("g4_meta.npz")
    g = load_golden("g5_batched.npz")
    o = make_oracle(meta, "knn")
    o.max_ADMM_iter, o.ADMM_tol = 40, tol
    ref = o.combined_loop(g["y"][:B])
    n_ref = len(o.hist.p_res_list)
    assert 1 < n_ref < 40
    for r in range(world):
        assert int(np.load(tmp_path / f"gn_rank{r}.npy")[0]) == n_ref
        np.testing.assert_allclose(np.load(tmp_path / f"gx_rank{r}.npy"), ref, rtol=1e-12)
