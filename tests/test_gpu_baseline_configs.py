"""Every BASELINE.json configuration at its FULL size on the GPU, anchored to the CPU oracle (itself pinned to the
reference by tests/test_oracle_golden.py).

The oracle cannot run thousands of windows in seconds, and it does not have to: samples are independent
optimisations, so the GPU solves the whole production batch -- production kernel geometry: vector
width, column chunks, tile size, halo lists, cluster order, workspace size -- and the oracle solves a handful of the
SAME windows.  The comparison uses the per-sample history of the C ABI (mgadmm_history.metrics_per_sample), from
which the oracle's whole-batch norms over its own windows are re-formed.

Tolerances (float32 kernels vs the float64 oracle): ||x - x_ref|| / ||x_ref|| <= 1e-5 per sample,
residual history rel <= 1e-3 (absolute floor 1e-7 * ||x_ref|| = 2 ulp of float32 for the norms of differences),
CG counts within +-1.

  cfg2  PEMS04-shaped N=307, B=4096, LDS-resident path        16 windows x 5 iterations
  cfg3  10k-node kNN graph, B=512, LDS-tiled streaming path    4 windows x 2 iterations
  cfg4  100k-node mixed graph, B=256 (one GPU's slice of 2048) 2 windows x 2 iterations + operator properties
  cfg5  cfg2 graph, B=4096, both float32 paths                 64 windows x 50 iterations (tolerance sweep)
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

X_TOL, HIST_RTOL, CG_SLACK = 1e-5, 1e-3, 1


def _bench():
    import bench
    return bench


def _problem(workload, **kw):
    import mgadmm
    b = _bench()
    n, B, cl, dl, info, _ = b.build_problem(workload)
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl),
                                record_cg_coeffs=False, **kw)
    return blk, n, B, cl, info


def _oracle(blk, cl, info):
    from oracle import admm_oracle as orc
    return orc.OracleADMM(cl.numpy(), blk.u_ew[0].numpy(), blk.d_ew[0].numpy(), info, mode="knn", t_in=12, T=24)


def _check_windows(tag, blk, x, idx, o, xo, xtol=X_TOL, htol=HIST_RTOL, slack=CG_SLACK):
    """GPU batch result `x` (+ blk.metrics_per_sample, blk.CG_iter_*) against the oracle run on windows `idx`."""
    from mgadmm import _lib as L
    idx = np.asarray(idx)
    xg = x[torch.as_tensor(idx, device=x.device)].double().cpu().numpy()
    err = np.linalg.norm((xg - xo).reshape(len(idx), -1), axis=1) / np.linalg.norm(xo.reshape(len(idx), -1), axis=1)
    assert err.max() < xtol, (tag, "x", err.max())
    mps = blk.metrics_per_sample[:, :, idx]                  # (iters, NMETRIC, k) per-sample sums
    h = o.hist
    n_it = len(h.p_res_list)
    assert mps.shape[0] == n_it
    # norms of DIFFERENCES of float32 vectors (||x - x_old||, ||z - z_old||, ...) carry the rounding of the vectors
    # themselves: 2 ulp of float32 relative to ||x_ref|| is the resolution (cfg4, iteration 0: ||x1 - x0|| = 1.4 on
    # ||x|| = 5.4e5 -- the initial guess almost solves the first x-update -- measured difference 0.03 = 6e-8 ||x||)
    floor = 1e-7 * float(np.linalg.norm(xo))
    norm = lambda m: np.sqrt(mps[:, m].sum(1))
    mean = lambda m: mps[:, m].mean(1)
    pri = np.stack([norm(L.M_PRI_ZU), norm(L.M_PRI_PHI), norm(L.M_PRI_ZD)], 1)
    dual = np.stack([norm(L.M_DUAL_ZU), norm(L.M_DUAL_PHI), norm(L.M_DUAL_ZD)], 1)
    np.testing.assert_allclose(pri, np.array(h.p_res_list), rtol=htol, atol=floor, err_msg=f"{tag} primal residuals")
    np.testing.assert_allclose(dual, np.array(h.d_res_list), rtol=htol, atol=floor, err_msg=f"{tag} dual residuals")
    np.testing.assert_allclose(norm(L.M_XSHIFT), np.array(h.x_shift_list), rtol=htol, atol=floor, err_msg=f"{tag} x shift")
    np.testing.assert_allclose(norm(L.M_RECOVER), np.array(h.recover_list), rtol=htol, atol=floor, err_msg=f"{tag} ||Hx-y||")
    np.testing.assert_allclose(mean(L.M_GLR), np.array(h.GLR_list), rtol=htol, err_msg=f"{tag} GLR")
    np.testing.assert_allclose(mean(L.M_DGTV), np.array(h.DGTV_list), rtol=htol, err_msg=f"{tag} DGTV")
    np.testing.assert_allclose(mean(L.M_DGLR), np.array(h.DGLR_list), rtol=htol, err_msg=f"{tag} DGLR")
    for nm in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd"):
        got = torch.stack(getattr(blk, nm)).numpy()[:, idx]
        ref = np.array(getattr(h, nm)).reshape(n_it, -1)
        assert (got > 0).all(), (tag, nm, "CG did not converge")
        assert np.abs(got - ref).max() <= slack, (tag, nm, np.abs(got - ref).max())


def _solve(blk, y, iters):
    blk.max_ADMM_iter = iters
    blk.check_stop = False
    blk._reset_history()
    return blk.solve(y, print_info=False, return_state=False, per_sample_history=True)[0]


# ---------------------------------------------------------------------------------------------- cfg2 / cfg5
def test_cfg2_full_batch_lds_path_vs_oracle_windows():
    b = _bench()
    blk, n, B, cl, info = _problem("cfg2", path="lds")
    assert B == 4096
    y = b.synth_y(n, B, 12, seed=1, offset=0, device=torch.device("cuda"))
    x = _solve(blk, y, 5)
    from mgadmm import _lib
    assert _lib.lib.mgadmm_solver_path(blk._solvers[(1, torch.float32)][0], B) == _lib.PATH_LDS
    idx = np.linspace(0, B - 1, 16).astype(int)
    o = _oracle(blk, cl, info)
    xo = o.combined_loop(y[torch.as_tensor(idx, device=y.device)].double().cpu().numpy(), n_iters=5)
    _check_windows("cfg2-lds", blk, x, idx, o, xo)
    blk.close()


# ---------------------------------------------------------------------------------------------- sizes around the path switch
@pytest.mark.parametrize("n,path,tpg", [(341, "lds", 8), (358, "lds", 12), (512, "lds", 12), (600, "stream", 0), (883, "stream", 0)])
def test_sizes_around_the_path_switch_vs_oracle_windows(n, path, tpg):
    """PEMS-like graphs of 341 ... 883 nodes (PEMS03 = 358, PEMS07 = 883) at B = 512, the host's own choice of path and node
    order: the last size of the 8-step LDS instance, the 12-step instances up to their limit of 512 nodes, and the streaming
    path just beyond it (cluster order, fused Ldr^T Ldr kernel in its 12-, 16- or 24-slot instance) -- 4 windows x 4 iterations
    against the oracle with the tolerances of the BASELINE configs."""
    import math
    import mgadmm
    from mgadmm import _lib, gpu_graph
    b = _bench()
    dev = torch.device("cuda")
    ue, ud = b.pems_like_graph(n, int(round(n * 1.11)), seed=0)
    cl, dl = gpu_graph.k_nearest_neighbors(n, ue, ud, 4, device=dev)
    cl = cl.to(torch.int64).cpu()
    r = math.sqrt(n / 24)
    info = dict(rho=2 * r, rho_u=3 * r, rho_d=2 * r, mu_u=1, mu_d1=2, mu_d2=1)
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl.cpu()),
                                record_cg_coeffs=False)
    B = 512
    y = b.synth_y(n, B, 12, seed=1, offset=0, device=dev)
    x = _solve(blk, y, 4)
    h = blk._solvers[(1, torch.float32)][0]
    assert _lib.lib.mgadmm_solver_path(h, B) == (_lib.PATH_LDS if path == "lds" else _lib.PATH_STREAM)
    if path == "lds":
        assert _lib.query(h, _lib.Q_LDS_TPG) == tpg
    else:
        assert _lib.query(h, _lib.Q_TILE_ROWS) == 8 and _lib.query(h, _lib.Q_CLDR_SLOTS) in (12, 16, 24)
    idx = np.array([0, 170, 341, B - 1])
    o = _oracle(blk, cl, info)
    xo = o.combined_loop(y[torch.as_tensor(idx, device=y.device)].double().cpu().numpy(), n_iters=4)
    _check_windows(f"n{n}-{path}", blk, x, idx, o, xo)
    blk.close()


_CFG5_ORACLE = {}


@pytest.mark.parametrize("path", ["lds", "stream"])
def test_cfg5_fp32_paths_vs_fp64_oracle_tolerance_sweep(path):
    """BASELINE config 5: PEMS04 graph, 'None' ablation (asymmetric L_d), B = 4096 in float32 on the GPU against the
    float64 CPU oracle on the first 64 windows, 50 iterations, every iteration's residuals and CG counts."""
    b = _bench()
    blk, n, B, cl, info = _problem("cfg2", path=path)
    y = b.synth_y(n, B, 12, seed=1, offset=0, device=torch.device("cuda"))
    x = _solve(blk, y, 50)
    idx = np.arange(64)
    if "o" not in _CFG5_ORACLE:                     # ~20 s of CPU: shared by the two GPU paths
        o = _oracle(blk, cl, info)
        _CFG5_ORACLE["o"] = (o, o.combined_loop(y[:64].double().cpu().numpy(), n_iters=50))
    o, xo = _CFG5_ORACLE["o"]
    _check_windows(f"cfg5-{path}", blk, x, idx, o, xo)
    blk.close()


# ---------------------------------------------------------------------------------------------- cfg3
def test_cfg3_production_tile_geometry_vs_oracle_windows():
    """B = 512 (VEC = 4, two column chunks), 8-row LDS tiles with 20-row halos on the real 10k-node kNN graph in
    cluster order -- the geometry the roofline figure is measured on."""
    b = _bench()
    blk, n, B, cl, info = _problem("cfg3")
    assert (n, B) == (10000, 512)
    y = b.synth_y(n, B, 12, seed=1, offset=0, device=torch.device("cuda"))
    x = _solve(blk, y, 2)
    idx = np.array([0, 170, 341, 511])
    o = _oracle(blk, cl, info)
    xo = o.combined_loop(y[torch.as_tensor(idx, device=y.device)].double().cpu().numpy(), n_iters=2)
    _check_windows("cfg3", blk, x, idx, o, xo)
    blk.close()


# ---------------------------------------------------------------------------------------------- cfg4
@pytest.fixture(scope="module")
def cfg4():
    blk, n, B, cl, info = _problem("cfg4", bug_compat=False)
    assert (n, B) == (100000, 256)
    yield blk, n, B, cl, info
    blk.close()


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def test_cfg4_operators_linear_adjoint_and_tile_vs_plain(cfg4, monkeypatch):
    """One GPU's slice of BASELINE config 4: N = 100 000, B = 256 (12.5k tiles per operator, one column chunk)."""
    blk, n, B, cl, info = cfg4
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 24, n, 1, device="cuda", generator=g)
    y = torch.randn(B, 24, n, 1, device="cuda", generator=g)
    outs = {}
    for nm in ("Lu", "Ldr", "Ldr_T", "cLdr"):
        op = getattr(blk, "apply_op_" + nm)
        ox = outs[nm] = op(x)
        oy = op(y)
        lin = op(2.0 * x - 0.5 * y)
        assert _rel(lin, 2.0 * ox - 0.5 * oy) < 2e-6, nm
        del lin, oy
    # exact transpose with the quirk off, per sample
    a = (outs["Ldr"] * y).sum((1, 2, 3)).double()
    bb = (x * blk.apply_op_Ldr_T(y)).sum((1, 2, 3)).double()
    assert float(((a - bb).abs() / (a.abs() + 1e4)).max()) < 1e-4
    assert float((x * outs["cLdr"]).sum((1, 2, 3)).min()) >= 0.0
    del y
    # the LDS-tiled kernel against the plain row kernel (independent code paths) on the same vectors
    monkeypatch.setenv("MGADMM_TILE", "0")
    plain, _, _, _, _ = _problem("cfg4", bug_compat=False)
    for nm in ("Lu", "Ldr", "Ldr_T", "cLdr"):
        assert _rel(outs[nm], getattr(plain, "apply_op_" + nm)(x)) < 1e-6, nm
    outs.clear()
    for nm in ("LHS_x", "LHS_zu", "LHS_zd"):
        assert _rel(getattr(blk, nm)(x), getattr(plain, nm)(x)) < 1e-6, nm
    plain.close()


def test_cfg4_solve_finite_repeatable_and_vs_oracle_windows(cfg4):
    blk, n, B, cl, info = cfg4
    b = _bench()
    y = b.synth_y(n, B, 12, seed=1, offset=0, device=torch.device("cuda"))
    x1 = _solve(blk, y, 2)
    h1 = np.array(blk.p_res_list)
    assert torch.isfinite(x1).all() and np.isfinite(h1).all()
    idx = np.array([0, B - 1])
    from oracle import admm_oracle as orc
    o = orc.OracleADMM(cl.numpy(), blk.u_ew[0].numpy(), blk.d_ew[0].numpy(), info, mode="knn", t_in=12, T=24,
                       bug_compat=False)
    xo = o.combined_loop(y[torch.as_tensor(idx, device=y.device)].double().cpu().numpy(), n_iters=2)
    _check_windows("cfg4", blk, x1, idx, o, xo)
    x2 = _solve(blk, y, 2)
    assert torch.equal(x1, x2) and np.array_equal(h1, np.array(blk.p_res_list))      # bitwise repeatable
    ws = blk.workspace_bytes()
    assert 40e9 < ws < 80e9, ws                                                       # ~52 GB of the 288 GB


def test_cfg4_default_quirks_vs_oracle_windows():
    """The same slice with the reference's quirks ON (bug_compat=True, the default of both sides: Q1 keeps the identity on
    the t = 0 block of Ldr_T, ADMM.py:221-222, which enters RHS_x): 2 windows x 2 iterations against the oracle."""
    blk, n, B, cl, info = _problem("cfg4")
    b = _bench()
    y = b.synth_y(n, B, 12, seed=1, offset=0, device=torch.device("cuda"))
    x = _solve(blk, y, 2)
    idx = np.array([0, B - 1])
    o = _oracle(blk, cl, info)
    xo = o.combined_loop(y[torch.as_tensor(idx, device=y.device)].double().cpu().numpy(), n_iters=2)
    _check_windows("cfg4-quirks", blk, x, idx, o, xo)
    blk.close()
