"""The three schedules of the LDS path's ADMM outer loop (csrc/engine.h, solve_lds) give the same bits.

SYNC (MGADMM_LDS_ASYNC=0): one stream, the host tests the stop criterion (ADMM.py:645-646) after every iteration.
DEVSTOP (check_stop): the stop test runs on the device, later launches return at their guard, the host looks late.
CHUNKS (fixed iteration count): one k_admm_lds launch runs J iterations on every sample (MGADMM_LDS_CHUNK, default 7; the
workgroup keeps its sample), every iterate goes to a buffer of its own and the whole-batch metric kernels of a chunk run on a
helper stream beside the launch of the next chunk.
Compared: x, the exported state, every history list and the iteration count -- bit for bit -- plus the stop iteration against
the oracle (float64 restatement of the reference)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import case_inputs, make_oracle, make_product

pytestmark = pytest.mark.gpu

LISTS = ("p_res_list", "d_res_list", "x_shift_list", "GLR_list", "DGTV_list", "DGLR_list", "recover_list")


def _run(env, meta, abl, task, B, check_stop, iters, tol=None, chunk=None):
    old = os.environ.get("MGADMM_LDS_ASYNC")
    os.environ["MGADMM_LDS_ASYNC"] = env          # read when the solver handle is created
    if chunk is not None:
        os.environ["MGADMM_LDS_CHUNK"] = str(chunk)
    try:
        y, mask = case_inputs(meta, task, np.float32)
        rng = np.random.default_rng(5)
        reps = -(-B // y.shape[0])
        y = np.concatenate([y * (1 + 0.01 * r) for r in range(reps)])[:B] + (rng.random((B,) + y.shape[1:]) * 0.5).astype(np.float32)
        if mask is not None:
            mask = np.concatenate([mask] * reps)[:B]
            y = y * mask
        blk = make_product(meta, "knn", ablation=abl, path="lds")
        blk.check_stop = check_stop
        blk.max_ADMM_iter = iters
        if tol is not None:
            blk.ADMM_tol = tol
        x, (zu, zd), phi, _ = blk.solve(torch.from_numpy(y), mask=torch.from_numpy(mask) if mask is not None else None)
        out = dict(x=x.clone(), zu=zu.clone(), zd=None if zd is None else zd.clone(), phi=None if phi is None else phi.clone(),
                   state={k: v.clone() for k, v in blk.state.items() if v is not None},
                   lists={k: np.array(getattr(blk, k), dtype=np.float64) for k in LISTS if len(getattr(blk, k))},
                   dxps=np.array([np.asarray(v) for v in blk.delta_x_per_step]),
                   cg=[[torch.as_tensor(v).clone() for v in getattr(blk, nm)] for nm in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd")],
                   n=len(blk.p_res_list), y=y, mask=mask)
        blk.close()
        return out
    finally:
        os.environ.pop("MGADMM_LDS_CHUNK", None)
        if old is None:
            os.environ.pop("MGADMM_LDS_ASYNC", None)
        else:
            os.environ["MGADMM_LDS_ASYNC"] = old


def _same(a, b):
    assert a["n"] == b["n"]
    for k in ("x", "zu", "zd", "phi"):
        assert (a[k] is None) == (b[k] is None)
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), k
    assert a["state"].keys() == b["state"].keys()
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), "state." + k
    assert a["lists"].keys() == b["lists"].keys()
    for k in a["lists"]:
        np.testing.assert_array_equal(a["lists"][k], b["lists"][k], err_msg=k)
    np.testing.assert_array_equal(a["dxps"], b["dxps"])
    for la, lb in zip(a["cg"], b["cg"]):
        assert len(la) == len(lb) and all(torch.equal(u, v) for u, v in zip(la, lb))


@pytest.mark.parametrize("abl,task,B", [("None", "pred", 3), ("None", "pred", 700), ("DGLR", "mask", 70), ("DGTV", "pred", 130)])
def test_fixed_iteration_count_overlapped_equals_synchronous(abl, task, B):
    meta = load_golden("g4_meta.npz")
    for iters in (1, 2, 3, 7):          # the x buffer holding the result changes with max_iter mod 3 (mod 2: synchronous)
        a = _run("0", meta, abl, task, B, False, iters)
        b = _run("1", meta, abl, task, B, False, iters)
        assert a["n"] == iters
        _same(a, b)


@pytest.mark.parametrize("abl,task,B", [("None", "pred", 3), ("None", "pred", 300), ("DGLR", "mask", 70), ("DGTV", "pred", 130)])
def test_several_iterations_per_launch_equal_one_per_launch(abl, task, B):
    """Chunks of J iterations per launch (the workgroup keeps its sample, iterates in a ring of 2 (J - 1) + 3 buffers) against the
    synchronous one-iteration-per-launch loop: iteration counts that are a multiple of J, leave a remainder, fit one chunk,
    and need more chunks than there are buffer sets."""
    meta = load_golden("g4_meta.npz")
    for iters, chunk in ((1, 7), (5, 2), (6, 3), (7, 7), (8, 7), (16, 5), (23, 4), (20, 10), (37, 16), (50, 12)):     # chunks > 7: iterate buffers beyond the workspace vectors
        a = _run("0", meta, abl, task, B, False, iters)
        b = _run("1", meta, abl, task, B, False, iters, chunk=chunk)
        assert a["n"] == b["n"] == iters
        _same(a, b)


@pytest.mark.parametrize("abl,task,B,stop_at", [("None", "pred", 3, 9), ("None", "pred", 300, 4), ("DGLR", "mask", 70, 13),
                                                 ("None", "pred", 3, None), ("DGTV", "pred", 130, 23)])
def test_device_side_stop_equals_host_side_stop(abl, task, B, stop_at):
    meta = load_golden("g4_meta.npz")
    if stop_at is None:
        tol = 1e-30                               # never reached: all 25 iterations run
    else:                                         # a tolerance the residual maxima fall below near iteration `stop_at`
        free = _run("0", meta, abl, task, B, False, 25)
        worst = np.maximum(free["lists"]["p_res_list"].max(axis=1), free["lists"]["d_res_list"].max(axis=1))
        tol = float(worst[stop_at]) * 1.0001
    a = _run("0", meta, abl, task, B, True, 25, tol)
    b = _run("1", meta, abl, task, B, True, 25, tol)
    if stop_at is None:
        assert a["n"] == 25
    else:
        assert 1 < a["n"] <= stop_at + 1, a["n"]  # the stop test fires inside the loop (speculative launches follow it)
    _same(a, b)


def test_device_side_stop_iteration_matches_the_oracle():
    meta = load_golden("g4_meta.npz")
    y, _ = case_inputs(meta, "pred", np.float64)
    o = make_oracle(meta, "knn")
    o.ADMM_tol, o.max_ADMM_iter = 40.0, 60
    o.combined_loop(y)
    n_ref = len(o.hist.p_res_list)
    blk = make_product(meta, "knn", path="lds")
    blk.ADMM_tol, blk.max_ADMM_iter = 40.0, 60
    blk.combined_loop(torch.from_numpy(y.astype(np.float32)), print_info=False)
    assert len(blk.p_res_list) == n_ref
    blk.close()


def test_nonfinite_input_is_reported_by_both_schedules():
    meta = load_golden("g4_meta.npz")
    y, _ = case_inputs(meta, "pred", np.float32)
    y = y.copy()
    y[0, 3, 4, 0] = np.nan
    for env in ("0", "1"):
        for cs in (True, False):
            os.environ["MGADMM_LDS_ASYNC"] = env
            try:
                blk = make_product(meta, "knn", path="lds")
                blk.check_stop = cs
                blk.max_ADMM_iter = 6
                with pytest.raises(AssertionError):
                    blk.combined_loop(torch.from_numpy(y), print_info=False)
                blk.close()
            finally:
                os.environ.pop("MGADMM_LDS_ASYNC", None)


def test_overlapped_schedule_on_a_side_stream_of_the_caller():
    """The ABI takes the caller's stream: the helper stream of the overlapped schedule is ordered against it with events, also
    when it is not the default stream; the result is bitwise the default-stream result."""
    meta = load_golden("g4_meta.npz")
    y, _ = case_inputs(meta, "pred", np.float32)
    y = np.concatenate([y * (1 + 0.01 * r) for r in range(100)])
    blk = make_product(meta, "knn", path="lds")
    blk.check_stop = False
    blk.max_ADMM_iter = 8
    yt = torch.from_numpy(y).cuda()
    x0 = blk.combined_loop(yt, print_info=False).clone()
    h0 = np.array(blk.p_res_list)
    blk._reset_history()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x1 = blk.combined_loop(yt, print_info=False)
    s.synchronize()
    assert torch.equal(x0, x1)
    np.testing.assert_array_equal(h0, np.array(blk.p_res_list))
    blk.close()
