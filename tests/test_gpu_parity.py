"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden vectors the
reference produced.  Tolerances:
  float64 kernels vs reference float64:  rel <= 1e-10 on vectors, identical CG counts
  float32 kernels vs reference float64:  ||x - x_ref|| / ||x_ref|| <= 1e-5, history rel <= 1e-3,
                                         CG counts within +-1 (+-2 on the 2-iteration diagonal solves)
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import case_inputs, make_oracle, make_product, meta_from_g2, rel

pytestmark = pytest.mark.gpu

F32_X_TOL = 1e-5
F32_HIST_RTOL = 1e-3


def T_(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


# ------------------------------------------------------------------ graph handle
def test_transpose_built_in_library_is_exact():
    import scipy.sparse as sp
    from mgadmm.graph import Graph, tables_to_csr
    g = load_golden("g4_meta.npz")
    n = int(g["n"])
    u = tables_to_csr(g["knn_cl"], g["knn_u_ew"], 1)
    d = tables_to_csr(g["knn_cl"], g["knn_d_ew"], 0)
    for reorder in (False, True):
        gr = Graph(n, 24, u, d, reorder=reorder)
        rp, col, val = gr.transpose_csr()
        Wd = sp.csr_matrix((d[2], d[1], d[0]), shape=(n, n))
        WdT = sp.csr_matrix((val, col, rp), shape=(n, n))
        assert abs(WdT - Wd.T).max() == 0
        perm = gr.perm()
        assert sorted(perm.tolist()) == list(range(n))
        gr.close()


# ------------------------------------------------------------------ operators vs golden dense matrices (G2)
@pytest.mark.parametrize("mode", ["knn", "physical", "line", "skip3"])
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_operators_match_reference_dense(mode, dt):
    g = load_golden(f"g2_ops_{mode}.npz")
    meta = meta_from_g2(g)
    T, n = int(g["T"]), int(g["n"])
    blk = make_product(meta, mode, compute_dtype=dt)
    size = T * n
    eye = torch.eye(size, dtype=dt).reshape(size, T, n, 1)          # batch of 72 unit vectors: B=72 -> VEC 1
    tol = 1e-13 if dt == torch.float64 else 2e-6
    for nm in ("Lu", "Ldr", "Ldr_T", "cLdr"):
        out = getattr(blk, "apply_op_" + nm)(eye)
        assert out.dtype == dt and out.shape == eye.shape and out.device == eye.device
        D = out.reshape(size, size).T.double().numpy()
        assert rel(D, g[nm]) < tol, (mode, nm)
    assert rel(blk.LHS_zu(eye).reshape(size, size).T, g["LHS_zu"]) < tol
    assert rel(blk.LHS_zd(eye).reshape(size, size).T, g["LHS_zd"]) < tol
    m = T_(g["mask"], dt).expand(size, -1, -1, -1).contiguous()
    assert rel(blk.LHS_x(eye, mask=m).reshape(size, size).T, g["LHS_x_mask_None"]) < tol
    for abl in ("None", "DGLR", "DGTV"):
        b2 = make_product(meta, mode, ablation=abl, compute_dtype=dt)
        assert rel(b2.LHS_x(eye).reshape(size, size).T, g["LHS_x_" + abl]) < tol, abl
        b2.close()
    blk.close()


@pytest.mark.parametrize("B", [1, 3, 100, 200, 260])
def test_operators_all_vector_widths_and_reorder(B, monkeypatch):
    """B = 1..260 walks the VEC=1/2/4 kernels and ragged column padding; reorder=True exercises the
    internal node permutation.  Checked against the oracle on random data."""
    meta = load_golden("g4_meta.npz")
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, 24, 30, 1))
    if B in (1, 200):
        monkeypatch.setenv("MGADMM_TILE", "0")          # "cluster" order with the plain row kernel (default: LDS-tiled)
    for mode in ("knn", "skip3"):
        o = make_oracle(meta, mode)
        for reorder in (False, "rcm", "cluster"):          # "cluster" also switches the spatial ops to the LDS-tiled kernel
            blk = make_product(meta, mode, compute_dtype=torch.float64, reorder=reorder)
            for nm in ("Lu", "Ldr", "Ldr_T", "cLdr"):
                got = getattr(blk, "apply_op_" + nm)(T_(x))
                assert rel(got, getattr(o, "apply_op_" + nm)(x)) < 1e-13, (mode, nm, reorder)
            blk.close()
    blk = make_product(meta, "knn", compute_dtype=torch.float32)
    assert rel(blk.apply_op_cLdr(T_(x, torch.float32)), make_oracle(meta, "knn").apply_op_cLdr(x)) < 2e-6
    blk.close()


def test_bug_compat_switch_q1():
    meta = load_golden("g4_meta.npz")
    x = np.random.default_rng(0).standard_normal((2, 24, 30, 1))
    for bc in (True, False):
        blk = make_product(meta, "knn", compute_dtype=torch.float64, bug_compat=bc)
        o = make_oracle(meta, "knn", bug_compat=bc)
        assert rel(blk.apply_op_Ldr_T(T_(x)), o.apply_op_Ldr_T(x)) < 1e-13
        blk.close()
    # exact transpose when the quirk is off: <Ldr x, y> == <x, Ldr_T y>
    blk = make_product(meta, "knn", compute_dtype=torch.float64, bug_compat=False)
    y = T_(np.random.default_rng(1).standard_normal((2, 24, 30, 1)))
    xt = T_(x)
    lhs = (blk.apply_op_Ldr(xt) * y).sum().item()
    rhs = (xt * blk.apply_op_Ldr_T(y)).sum().item()
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))
    blk.close()


def test_channels_fold_into_nodes():
    meta = load_golden("g4_meta.npz")
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 24, 30, 2))
    o = make_oracle(meta, "knn")
    blk = make_product(meta, "knn", compute_dtype=torch.float64)
    for nm in ("Lu", "Ldr", "Ldr_T", "cLdr"):
        assert rel(getattr(blk, "apply_op_" + nm)(T_(x)), getattr(o, "apply_op_" + nm)(x)) < 1e-13
    assert rel(blk.LHS_x(T_(x)), o.LHS_x(x)) < 1e-13
    blk.close()


@pytest.mark.parametrize("mode,task", [("knn", "pred"), ("knn", "mask"), ("skip3", "pred")])
def test_multi_channel_full_solve(mode, task):
    """C = 3 channels: one CG system per sample over (T, N, C) (alpha, beta and the convergence test are shared
    by the channels, ADMM.py:347-360).  The product folds channels into the node axis; float64 kernels vs the
    oracle, and the float32 paths (the folded graph has 90 nodes: LDS path at TPG = 3)."""
    meta = load_golden("g4_meta.npz")
    rng = np.random.default_rng(11)
    B, C_ = 3, 3
    x_true = 100 + 50 * rng.standard_normal((B, 24, 30, C_))
    if task == "pred":
        y, mask = x_true[:, :12].copy(), None
    else:
        mask = (rng.random(x_true.shape) >= 0.4).astype(np.float32)
        y = x_true * mask
    o = make_oracle(meta, mode)
    xo = o.combined_loop(y, mask=mask, n_iters=5)
    for kw, xtol, htol, slack in ((dict(compute_dtype=torch.float64), 1e-10, 1e-8, 0), (dict(path="lds"), 1e-5, 1e-3, 1),
                                  (dict(path="stream"), 1e-5, 1e-3, 1)):
        blk = make_product(meta, mode, **kw)
        blk.max_ADMM_iter = 5
        blk.check_stop = False
        # float32 mask like utils.py:129 (the reference's time moments are float32 then)
        x = blk.combined_loop(T_(y), mask=torch.from_numpy(mask) if mask is not None else None, print_info=False)
        assert tuple(x.shape) == (B, 24, 30, C_) and rel(x, xo) < xtol, kw
        floor = (1e-8 if htol >= 1e-4 else 1e-14) * float(np.linalg.norm(xo))
        np.testing.assert_allclose(np.array(blk.p_res_list), np.array(o.hist.p_res_list), rtol=htol, atol=floor)
        got = torch.stack(blk.CG_iter_x).numpy()
        assert np.abs(got - np.array(o.hist.CG_iter_x)).max() <= slack
        blk.close()


# ------------------------------------------------------------------ KATs (G6)
def test_line_graph_kats_and_phi_direct():
    from mgadmm.ADMM import ADMM_algorithm
    g = load_golden("g6_kats.npz")
    x = T_(g["line_x"])
    info = {"rho": 2, "rho_u": 1, "rho_d": 1, "mu_u": 1, "mu_d1": 1, "mu_d2": 1}
    cl = torch.tensor([[0, 1], [1, 0], [2, 1]])
    dl = torch.tensor([[0.0, 1.0], [0.0, 1.0], [0.0, 2.0]])
    for skip in (1, 2):
        blk = ADMM_algorithm({"n_nodes": 3}, info, use_kNN=True, k=1, u_sigma=5, d_sigma=5, t_in=3, T=5,
                             use_line_graph=True, skip_connection=skip, tables=(cl, dl), compute_dtype='match')
        np.testing.assert_allclose(blk.apply_op_Ldr(x).numpy(), g[f"line{skip}_Ldr"], rtol=1e-15)
        np.testing.assert_allclose(blk.apply_op_Ldr_T(x).numpy(), g[f"line{skip}_LdrT"], rtol=1e-15)
        if skip == 1:
            assert blk.apply_op_Ldr(x)[0, :, 0, 0].tolist() == [0, 1, 1, 1, 1]
            assert blk.apply_op_Ldr_T(x)[0, :, 0, 0].tolist() == [-2, -1, -1, -1, 5]
            phi = blk.phi_direct(x, torch.zeros_like(x))
            assert phi[0, :, 0, 0].tolist() == [0, .5, .5, .5, .5]
            np.testing.assert_array_equal(phi.numpy(), g["line1_phi"])
        else:
            assert blk.apply_op_Ldr(x)[0, :, 0, 0].tolist() == [0, 1, 1.5, 1.5, 1.5]
            assert blk.apply_op_Ldr_T(x)[0, :, 0, 0].tolist() == [-3.5, -1.5, -1.5, 1.5, 5]
        blk.close()


def test_initial_guess_and_interpolation_kats():
    from mgadmm.ADMM import initial_guess, initial_interpolation
    g = load_golden("g6_kats.npz")
    x32 = initial_guess(torch.from_numpy(g["ig_y32"]), 3, 6)
    assert x32.dtype == torch.float32
    np.testing.assert_allclose(x32.numpy(), g["ig_x32"], rtol=3e-7)
    np.testing.assert_allclose(initial_guess(T_(g["ig_y32"]), 3, 6).numpy(), g["ig_x64"], rtol=1e-14)
    np.testing.assert_allclose(initial_guess(T_(g["ig_rand_y"]), 12, 24).numpy(), g["ig_rand_x"], rtol=1e-13)
    xi = initial_interpolation(T_(g["ii_y"]), torch.from_numpy(g["ii_mask"]))
    np.testing.assert_allclose(xi.numpy(), g["ii_x"], rtol=1e-11, atol=1e-9)
    xi32 = initial_interpolation(T_(g["ii_y"], torch.float32), torch.from_numpy(g["ii_mask"]))
    np.testing.assert_allclose(xi32.numpy(), g["ii_x32"], rtol=5e-5, atol=5e-3)


def test_cg_script_kat():
    from mgadmm.CG_script import conjugate_gradient
    A = np.array([[4.0, 1.0], [1.0, 3.0]])
    b = np.array([1.0, 2.0])
    x, it = conjugate_gradient(A, b)                     # reference CG_script.py:47-56
    assert it == 2
    np.testing.assert_allclose(x, [0.09090909, 0.63636364], atol=1e-8)
    rng = np.random.default_rng(0)
    M = rng.standard_normal((40, 40))
    A = M @ M.T + 40 * np.eye(40)
    b = rng.standard_normal(40)
    from oracle.admm_oracle import conjugate_gradient as ocg
    x, it = conjugate_gradient(A, b, tol=1e-10)
    xo, ito = ocg(A, b, tol=1e-10)
    assert it == ito and rel(x, xo) < 1e-10
    x, it = conjugate_gradient(A, b, tol=1e-30, max_iter=5)
    assert it == 5


# ------------------------------------------------------------------ CG (G3)
@pytest.mark.parametrize("mode", ["knn", "skip3"])
def test_cg_matches_reference(mode):
    g = load_golden("g3_cg.npz")
    meta = {k: g[k] for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2", "T", "t_in", "n")}
    meta["knn_cl"], meta["knn_u_ew"] = g[f"{mode}_cl"], g[f"{mode}_u_ew"]
    meta["knn_d_ew"] = g["knn_d_ew"] if mode == "knn" else None
    rhs, x0, m = g[f"{mode}_rhs"], g[f"{mode}_x0"], g[f"{mode}_mask"]
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        blk = make_product(meta, mode, compute_dtype=dt)
        for nm, fn, kw in (("x", blk.LHS_x, {}), ("xmask", blk.LHS_x, {"mask": T_(m, dt)}), ("zu", blk.LHS_zu, {}),
                           ("zd", blk.LHS_zd, {})):
            x, it, al, be = blk.CG_solver(fn, T_(rhs, dt), T_(x0, dt), **kw)
            pre = f"{mode}_{nm}_f64_"
            assert isinstance(it, int) and al.dtype == torch.float32
            if dt == torch.float64:
                assert it == int(g[pre + "iters"]), nm
                assert rel(x, g[pre + "x"]) < 1e-11
                np.testing.assert_allclose(al.numpy(), g[pre + "alpha"], rtol=2e-7)
                np.testing.assert_allclose(be.numpy(), g[pre + "beta"], rtol=2e-7, atol=1e-12)
            else:
                assert abs(it - int(g[pre + "iters"])) <= 1, nm
                assert rel(x, g[pre + "x"]) < 1e-5
                assert abs(it - int(g[f"{mode}_{nm}_f32_iters"])) <= 1
        if dt == torch.float64:
            x, it, _, _ = blk.CG_solver(blk.LHS_zu, T_(rhs))
            assert it == int(g[f"{mode}_zu_zero_iters"]) and rel(x, g[f"{mode}_zu_zero_x"]) < 1e-11
        blk.close()


def test_cg_per_sample_convergence_and_max_iter():
    """B samples = B independent runs: each sample stops at its own iteration (quirk Q6)."""
    meta = load_golden("g4_meta.npz")
    o = make_oracle(meta, "knn")
    rng = np.random.default_rng(5)
    rhs = rng.standard_normal((5, 24, 30, 1)) * np.array([1e-6, 1.0, 1e3, 1.0, 1e-3]).reshape(5, 1, 1, 1)
    x0 = rng.standard_normal((5, 24, 30, 1)) * 1e-3
    blk = make_product(meta, "knn", compute_dtype=torch.float64)
    x, it, al, be = blk.CG_solver(blk.LHS_zd, T_(rhs), T_(x0))
    xo, ito, alo, beo = o.CG_solver(o.LHS_zd, rhs, x0)
    assert it.tolist() == ito.tolist() and len(set(it.tolist())) > 1
    assert rel(x, xo) < 1e-11
    K = alo.shape[0]
    np.testing.assert_allclose(al.numpy()[:K], alo, rtol=1e-9, equal_nan=True)
    assert np.isnan(al.numpy()[K:]).all()
    blk.max_CG_iter = 3
    x, it, al, be = blk.CG_solver(blk.LHS_zd, T_(rhs), T_(x0))
    o.max_CG_iter = 3
    xo, ito, _, _ = o.CG_solver(o.LHS_zd, rhs, x0)
    assert it.tolist() == ito.tolist() == [-1] * 5 and rel(x, xo) < 1e-12
    x1, it1, a1, b1 = blk.CG_solver(blk.LHS_zd, T_(rhs[:1]), T_(x0[:1]))
    assert it1 == -1 and isinstance(a1, list) and len(a1) == 3       # reference returns lists when not converged
    blk.close()


# ------------------------------------------------------------------ full solves (G4, G5)
def check_solve(blk, key, G, abl, x, xtol, htol, it_slack):
    assert rel(x, G("x")) < xtol, key
    n = int(G("n_iters"))
    assert len(blk.p_res_list) == n
    # residuals are norms of differences of O(|x|) numbers: every element of x - z carries a rounding error of eps |x_i|, so
    # the norm has an absolute noise floor of eps ||x|| (float32: eps = 6e-8 -> 1e-7 ||x||, the floor of
    # test_gpu_baseline_configs.py; float64: 1e-14 ||x||).  Entries of a nearly converged residual sit below it.
    floor = (1e-7 if htol >= 1e-4 else 1e-14) * float(np.linalg.norm(G("x")))
    np.testing.assert_allclose(np.array(blk.p_res_list), G("p_res"), rtol=htol, atol=floor)
    np.testing.assert_allclose(np.array(blk.d_res_list), G("d_res"), rtol=htol, atol=floor)
    np.testing.assert_allclose(blk.x_shift_list, G("x_shift"), rtol=htol)
    # per-time-step norms of x - x_old: differences of O(100) float32 numbers, so the absolute error
    # scales with the largest entry, not with each (possibly tiny) entry
    np.testing.assert_allclose(torch.stack(blk.delta_x_per_step).numpy(), G("dxps"), rtol=htol,
                               atol=htol * max(1e-3, float(G("dxps").max())))
    np.testing.assert_allclose(torch.stack(blk.GLR_list).numpy(), G("GLR"), rtol=htol)
    np.testing.assert_allclose(blk.recover_list, G("recover"), rtol=htol, atol=htol * 1e-2)
    if abl in ("None", "DGLR"):
        np.testing.assert_allclose(torch.stack(blk.DGTV_list).numpy(), G("DGTV"), rtol=htol)
    else:
        assert blk.DGTV_list == []
    if abl != "DGLR":
        np.testing.assert_allclose(torch.stack(blk.DGLR_list).numpy(), G("DGLR"), rtol=htol)
        assert np.abs(np.array(blk.CG_iter_zd) - G("CG_iter_zd")).max() <= it_slack
    else:
        assert blk.CG_iter_zd == [] and blk.DGLR_list == []
    slack_x = it_slack if abl in ("None", "DGLR") else max(it_slack, 2 * (it_slack > 0))
    assert np.abs(np.array(blk.CG_iter_x) - G("CG_iter_x")).max() <= slack_x, key
    assert np.abs(np.array(blk.CG_iter_zu) - G("CG_iter_zu")).max() <= it_slack, key
    assert blk.res_name == (["zu"] + (["phi"] if abl in ("None", "DGLR") else []) + (["zd"] if abl != "DGLR" else []))
    assert torch.Tensor(blk.p_res_list).shape == (n, len(blk.res_name))     # what plot_residual reads


def run_case(meta, solves, key, compute_dtype, path="auto"):
    mode, abl, task, tag, iters = key.split("-")
    np_dt = np.float64 if tag == "f64" else np.float32
    y, mask = case_inputs(meta, task, np_dt)
    blk = make_product(meta, mode, ablation=abl, compute_dtype=compute_dtype, path=path)
    blk.max_ADMM_iter = int(iters)
    yt = torch.from_numpy(y)
    mt = torch.from_numpy(mask) if mask is not None else None      # float32 mask like utils.py:129
    x, (zu, zd), phi, hist = blk.solve(yt, mask=mt)
    assert x.dtype == yt.dtype and x.device == yt.device and x.shape == (1, 24, 30, 1)
    return blk, x, zu, zd, phi


def all_keys(solves, tag):
    return sorted({k.split("/")[0] for k in solves.files if k.split("/")[0].split("-")[3] == tag})


def test_full_solves_f64_kernels_vs_reference_f64(g4_meta, g4_solves):
    n = 0
    for key in all_keys(g4_solves, "f64"):
        abl = key.split("-")[1]
        blk, x, zu, zd, phi = run_case(g4_meta, g4_solves, key, torch.float64)
        G = lambda f: g4_solves[f"{key}/{f}"]
        check_solve(blk, key, G, abl, x, 1e-10, 1e-8, 0)
        assert rel(zu, G("zu")) < 1e-10
        if abl != "DGLR":
            assert rel(zd, G("zd")) < 1e-10
        if abl in ("None", "DGLR"):
            assert rel(phi, G("phi")) < 1e-9
            a0 = blk.alpha_x[0].numpy()
            np.testing.assert_allclose(a0, G("alpha_x")[0][: len(a0)], rtol=2e-7)
        blk.close()
        n += 1
    assert n >= 30


@pytest.mark.parametrize("path", ["lds", "stream"])
def test_full_solves_f32_kernels_vs_reference_f64(g4_meta, g4_solves, path):
    """The product's default arithmetic (float32 HIP) against the reference's float64 iterates,
    float64 inputs: the tolerance sweep of BASELINE config 5 at fixture size.  Both execution paths:
    the LDS-resident fused kernel (one workgroup per sample) and the streaming batch-innermost kernels."""
    n = 0
    for key in all_keys(g4_solves, "f64"):
        abl = key.split("-")[1]
        blk, x, zu, zd, phi = run_case(g4_meta, g4_solves, key, torch.float32, path)
        assert rel(zu, g4_solves[f"{key}/zu"]) < 1e-4
        G = lambda f: g4_solves[f"{key}/{f}"]
        check_solve(blk, key, G, abl, x, F32_X_TOL, F32_HIST_RTOL, 1)
        blk.close()
        n += 1
    assert n >= 30


@pytest.mark.parametrize("tpg", [2, 3, 4, 6, 8, 12])
def test_lds_kernel_every_time_group_width(g4_meta, g4_solves, tpg, monkeypatch):
    """The fused LDS kernel is instantiated per time-group width TPG (time steps per thread; the plan picks the
    smallest that fits 1024 threads: 1 for the N = 30 fixtures, 8 for PEMS04).  Force each width on the
    golden solves -- kNN (gathers with aligned windows + edge groups), line and skip-3 (band stencils),
    all ablations, prediction and mask mode."""
    from mgadmm import _lib
    monkeypatch.setenv("MGADMM_LDS_TPG", str(tpg))
    n = 0
    for key in all_keys(g4_solves, "f64"):
        mode, abl, task, tag, iters = key.split("-")
        if int(iters) != 5:
            continue
        blk, x, zu, zd, phi = run_case(g4_meta, g4_solves, key, torch.float32, "lds")
        assert _lib.lib.mgadmm_solver_path(blk._solvers[(1, torch.float32)][0], 1) == _lib.PATH_LDS
        G = lambda f: g4_solves[f"{key}/{f}"]
        check_solve(blk, key, G, abl, x, F32_X_TOL, F32_HIST_RTOL, 1)
        blk.close()
        n += 1
    assert n >= 12


def test_full_solves_f32_inputs(g4_meta, g4_solves):
    n = 0
    for key in all_keys(g4_solves, "f32"):
        abl = key.split("-")[1]
        blk, x, zu, zd, phi = run_case(g4_meta, g4_solves, key, torch.float32)
        assert x.dtype == torch.float32
        G = lambda f: g4_solves[f"{key}/{f}"]
        check_solve(blk, key, G, abl, x, F32_X_TOL, 2e-3, 2)
        blk.close()
        n += 1
    assert n >= 30


@pytest.mark.parametrize("dt,xtol,htol,slack,path", [(torch.float64, 1e-10, 1e-8, 0, "stream"),
                                                       (torch.float32, 1e-5, 1e-3, 1, "stream"),
                                                       (torch.float32, 1e-5, 1e-3, 1, "lds")])
def test_batched_equals_looped_reference_runs(dt, xtol, htol, slack, path):
    g = load_golden("g5_batched.npz")
    meta = load_golden("g4_meta.npz")
    blk = make_product(meta, "knn", compute_dtype=dt, path=path)
    blk.max_ADMM_iter = int(g["iters"])
    x, (zu, zd), phi, hist = blk.solve(T_(g["y"]), per_sample_history=True)
    assert rel(x, g["x"]) < xtol
    assert rel(zu, g["zu"].reshape(x.shape)) < xtol * 10 and rel(phi, g["phi"].reshape(x.shape)) < xtol * 100
    itx = torch.stack(blk.CG_iter_x).numpy()                     # (iters, B)
    assert np.abs(itx.T - g["CG_iter_x"]).max() <= slack
    assert np.abs(torch.stack(blk.CG_iter_zd).numpy().T - g["CG_iter_zd"]).max() <= slack
    np.testing.assert_allclose(np.array(blk.p_res_list), np.sqrt((g["p_res"] ** 2).sum(0)), rtol=htol)
    np.testing.assert_allclose(np.array(blk.d_res_list), np.sqrt((g["d_res"] ** 2).sum(0)), rtol=htol)
    np.testing.assert_allclose(torch.stack(blk.GLR_list).numpy(), g["GLR"].mean(0), rtol=htol)
    np.testing.assert_allclose(torch.stack(blk.DGTV_list).numpy(), g["DGTV"].mean(0), rtol=htol)
    # per-sample sums of squares == the B single-sample reference histories
    from mgadmm import _lib
    ps = blk.metrics_per_sample                                   # (iters, NMETRIC, B)
    np.testing.assert_allclose(np.sqrt(ps[:, _lib.M_PRI_ZU, :]).T, g["p_res"][:, :, 0], rtol=htol)
    np.testing.assert_allclose(np.sqrt(ps[:, _lib.M_XSHIFT, :]).T, g["x_shift"], rtol=htol)
    blk.close()


@pytest.mark.parametrize("path", ["lds", "stream"])
def test_solve_is_bitwise_repeatable_and_inputs_untouched(path):
    meta = load_golden("g4_meta.npz")
    g = load_golden("g5_batched.npz")
    blk = make_product(meta, "knn", path=path)
    blk.max_ADMM_iter = 5
    y = T_(g["y"], torch.float32)
    y0 = y.clone()
    a = blk.combined_loop(y, print_info=False)
    h1 = np.array(blk.p_res_list)
    blk._reset_history()
    b = blk.combined_loop(y, print_info=False)
    assert torch.equal(a, b) and np.array_equal(h1, np.array(blk.p_res_list))
    assert torch.equal(y, y0)
    blk.close()


def test_early_stop_and_attribute_assignment():
    """ADMM_tol stop test (ADMM.py:645-646) and reference-style attribute overrides."""
    meta = load_golden("g4_meta.npz")
    y, _ = case_inputs(meta, "pred", np.float64)
    o = make_oracle(meta, "knn")
    o.ADMM_tol, o.max_ADMM_iter = 40.0, 60
    xo = o.combined_loop(y)
    n_ref = len(o.hist.p_res_list)
    assert 1 < n_ref < 60
    blk = make_product(meta, "knn", compute_dtype=torch.float64)
    blk.ADMM_tol, blk.max_ADMM_iter = 40.0, 60
    x = blk.combined_loop(torch.from_numpy(y), print_info=False)
    assert len(blk.p_res_list) == n_ref and rel(x, xo) < 1e-10
    blk.close()


def test_errors_follow_the_reference_conventions():
    from mgadmm.ADMM import ADMM_algorithm
    meta = load_golden("g4_meta.npz")
    with pytest.raises(AssertionError):
        make_product(meta, "knn", ablation="bogus")
    blk = make_product(meta, "knn")
    y = torch.zeros(1, 12, 30, 1)
    y[0, 3, 4, 0] = float("nan")
    with pytest.raises(AssertionError):
        blk.combined_loop(y, print_info=False)
    with pytest.raises(ValueError):
        blk.combined_loop(torch.zeros(1, 12, 31, 1), print_info=False)
    with pytest.raises(NotImplementedError):
        blk.CG_solver(lambda v: v, torch.zeros(1, 24, 30, 1))
    with pytest.raises(AssertionError):
        blk.combined_loop(torch.zeros(1, 24, 30, 1), mask=torch.ones(1, 24, 30, 1), differential=True, print_info=False)
    blk.close()


def test_print_info_format(capsys):
    meta = load_golden("g4_meta.npz")
    y, _ = case_inputs(meta, "pred", np.float64)
    blk = make_product(meta, "knn", compute_dtype=torch.float64)
    blk.max_ADMM_iter = 2
    blk.combined_loop(torch.from_numpy(y), print_info=True)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 2 and out[0].startswith("ADMM iters 0: x_CG_iters ") and "pri_err = [" in out[0]
    blk.close()


def test_lds_path_selection_and_cg_coefficients():
    """path='auto' takes the LDS-resident kernel for PEMS-size float32 problems and the streaming kernels
    otherwise; both record the CG coefficients like alpha_x / beta_x of the reference (example.ipynb cell 10)."""
    from mgadmm import _lib
    meta = load_golden("g4_meta.npz")
    g4 = load_golden("g4_solves.npz")
    y, _ = case_inputs(meta, "pred", np.float64)
    res = {}
    for path in ("lds", "stream"):
        blk = make_product(meta, "knn", path=path)
        blk.max_ADMM_iter = 3
        blk.combined_loop(torch.from_numpy(y), print_info=False)
        h = blk._solvers[(1, torch.float32)][0]
        assert _lib.lib.mgadmm_solver_path(h, 1) == (_lib.PATH_LDS if path == "lds" else _lib.PATH_STREAM)
        res[path] = blk
        a0 = blk.alpha_x[0].numpy()
        ref = g4["knn-None-pred-f64-5/alpha_x"][0]
        n = min(len(a0), int(np.isfinite(ref).sum()))
        np.testing.assert_allclose(a0[:n - 1], ref[:n - 1], rtol=2e-3)
        assert len(blk.alpha_zd) == 3 and len(blk.beta_zu) == 3
    np.testing.assert_allclose(np.array(res["lds"].p_res_list), np.array(res["stream"].p_res_list), rtol=1e-4)
    for b in res.values():
        b.close()
    blk = make_product(meta, "knn", compute_dtype=torch.float64, path="lds")
    with pytest.raises(_lib.MgadmmError):
        blk.combined_loop(torch.from_numpy(y), print_info=False)
    blk.close()


@pytest.mark.parametrize("dt,xtol,htol,slack", [(torch.float64, 1e-10, 1e-8, 0), (torch.float32, 1e-5, 1e-3, 1)])
def test_tiled_kernel_full_solves(g4_meta, g4_solves, dt, xtol, htol, slack, monkeypatch):
    """Streaming path on a cluster-ordered graph: the spatial operators run in the LDS-tiled kernel (k_tile)
    and the node permutation is applied at the ABI.  Same golden solves, same tolerances."""
    monkeypatch.setenv("MGADMM_TILE", "1")
    n = 0
    for key in all_keys(g4_solves, "f64"):
        mode, abl, task, tag, iters = key.split("-")
        if mode not in ("knn", "physical") or int(iters) != 5:
            continue
        y, mask = case_inputs(g4_meta, task, np.float64)
        blk = make_product(g4_meta, mode, ablation=abl, compute_dtype=dt, path="stream", reorder="cluster")
        blk.max_ADMM_iter = 5
        x, (zu, zd), phi, hist = blk.solve(torch.from_numpy(y), mask=torch.from_numpy(mask) if mask is not None else None)
        G = lambda f: g4_solves[f"{key}/{f}"]
        check_solve(blk, key, G, abl, x, xtol, htol, slack)
        assert rel(zu, G("zu")) < xtol * 10
        perm = blk._graphs[1][0].perm()
        assert sorted(perm.tolist()) == list(range(30)) and perm.tolist() != list(range(30))
        blk.close()
        n += 1
    assert n >= 9
    # batched, ragged column padding, VEC=4 chunks
    g = load_golden("g5_batched.npz")
    y = np.concatenate([g["y"]] * 33, 0)[:260]
    a = make_product(g4_meta, "knn", compute_dtype=dt, path="stream", reorder="cluster")
    b = make_product(g4_meta, "knn", compute_dtype=dt, path="stream", reorder=False)
    a.max_ADMM_iter = b.max_ADMM_iter = 3
    xa = a.combined_loop(torch.from_numpy(y), print_info=False)
    xb = b.combined_loop(torch.from_numpy(y), print_info=False)
    assert rel(xa, xb) < (1e-12 if dt == torch.float64 else 1e-5)
    a.close(); b.close()


# ------------------------------------------------------------------ fused cLdr kernel (k_cldr): every tile geometry
@pytest.mark.parametrize("geom,B,fold", [(1, 256, 1), (1, 256, 0), (2, 70, 1), (2, 70, 0), (2, 256, 1), (3, 256, 1), (3, 256, 0), (4, 256, 1), (4, 256, 0)])
def test_fused_cldr_kernel_geometries_vs_oracle(geom, B, fold, monkeypatch):
    """k_cldr = Ldr^T Ldr in one pass with q recomputed on the tile halo, with and without the CG vector update folded
    into its loads (CldrSrcFold), in each of its tile geometries (256 / 64 columns per row): operators, one CG solve and a
    whole 4-iteration solve against the oracle on the 30-node golden graph (several tiles: clusters are cut at the caps)."""
    from mgadmm import _lib
    monkeypatch.setenv("MGADMM_CLDR_GEOM", str(geom))
    monkeypatch.setenv("MGADMM_FOLD", str(fold))
    meta = load_golden("g4_meta.npz")
    rng = np.random.default_rng(geom * 10 + fold)
    x = rng.standard_normal((B, 24, 30, 1)).astype(np.float32)
    o = make_oracle(meta, "knn")
    blk = make_product(meta, "knn", compute_dtype=torch.float32, path="stream", reorder="cluster")
    xt = torch.from_numpy(x)
    assert rel(blk.apply_op_cLdr(xt), o.apply_op_cLdr(x.astype(np.float64))) < 2e-6
    assert rel(blk.LHS_x(xt), o.LHS_x(x.astype(np.float64))) < 2e-6
    assert rel(blk.LHS_zd(xt), o.LHS_zd(x.astype(np.float64))) < 2e-6
    k = 6                                                   # the oracle solves a few of the B systems
    rhs = (rng.standard_normal((B, 24, 30, 1)) * np.logspace(-1, 1, B).reshape(B, 1, 1, 1)).astype(np.float32)
    for fn_p, fn_o in ((blk.LHS_x, o.LHS_x), (blk.LHS_zd, o.LHS_zd)):
        xs, it, _, _ = blk.CG_solver(fn_p, torch.from_numpy(rhs), xt)
        xo, ito, _, _ = o.CG_solver(fn_o, rhs[:k].astype(np.float64), x[:k].astype(np.float64))
        assert rel(xs[:k], xo) < 1e-5 and np.abs(it[:k].numpy() - ito).max() <= 1
    y = 100 + 50 * rng.random((B, 12, 30, 1)).astype(np.float32)
    blk.max_ADMM_iter = 4
    blk.check_stop = False
    xs = blk.solve(torch.from_numpy(y), per_sample_history=True, return_state=False)[0]
    xo = o.combined_loop(y[:k].astype(np.float64), n_iters=4)
    assert rel(xs[:k], xo) < 1e-5
    mps = blk.metrics_per_sample[:, :, :k]
    np.testing.assert_allclose(np.sqrt(mps[:, _lib.M_PRI_ZD].sum(1)), np.array(o.hist.p_res_list)[:, 2], rtol=1e-3)
    np.testing.assert_allclose(np.sqrt(mps[:, _lib.M_PRI_PHI].sum(1)), np.array(o.hist.p_res_list)[:, 1], rtol=1e-3)
    got = torch.stack(blk.CG_iter_x).numpy()[:, :k]
    assert np.abs(got - np.array(o.hist.CG_iter_x)).max() <= 1
    h = blk._solvers[(1, torch.float32)][0]
    assert _lib.query(h, _lib.Q_TILE_ROWS) == 8
    blk.close()
