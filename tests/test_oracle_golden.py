"""Pins the CPU oracle (oracle/admm_oracle.py) against golden vectors produced by running the
reference itself (tests/golden/gen_golden.py) and against the reference's literal KATs."""
import numpy as np
import pytest

from conftest import admm_info_from, load_golden
from oracle import admm_oracle as orc

RTOL64 = 1e-12


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


# ---------------------------------------------------------------- G6 literal KATs
def test_kat_cg_script():
    # CG_script.py:47-56
    A = np.array([[4.0, 1.0], [1.0, 3.0]])
    b = np.array([1.0, 2.0])
    x, it = orc.conjugate_gradient(A, b)
    assert it == 2
    np.testing.assert_allclose(x, [0.09090909, 0.63636364], atol=1e-8)
    g = load_golden("g6_kats.npz")
    np.testing.assert_allclose(x, g["cg_x"], rtol=1e-14)
    assert it == int(g["cg_iters"])


def test_kat_connect_list():
    # utils.py:297-300
    edges = np.array([[0, 1], [1, 2], [2, 3], [3, 2], [2, 1], [1, 0]])
    cl, dl = orc.connect_list(4, edges, np.array([1, 2, 3, 3, 2, 1]))
    assert cl.tolist() == [[0, 1, -1], [1, 0, 2], [2, 1, 3], [3, 2, -1]]
    inf = np.inf
    assert dl.tolist() == [[0, 1, inf], [0, 1, 2], [0, 2, 3], [0, 3, inf]]
    g = load_golden("g6_kats.npz")
    assert np.array_equal(cl, g["cl4_cl"]) and np.array_equal(dl, g["cl4_dl"])


def _line_oracle(skip, T=5, N=3):
    cl = np.array([[0, 1], [1, 0], [2, 1]])
    info = {"rho": 2, "rho_u": 1, "rho_d": 1, "mu_u": 1, "mu_d1": 1, "mu_d2": 1}
    return orc.OracleADMM(cl, np.zeros((N, 1), np.float32), None, info, mode="line", t_in=3, T=T,
                          skip_connection=skip)


def test_kat_line_graph_operators():
    # directed_graph.ipynb cells 3-12 + SURVEY 8c G6
    g = load_golden("g6_kats.npz")
    x = g["line_x"]
    o1 = _line_oracle(1)
    assert o1.apply_op_Ldr(x)[0, :, 0, 0].tolist() == [0, 1, 1, 1, 1]
    assert o1.apply_op_Ldr_T(x)[0, :, 0, 0].tolist() == [-2, -1, -1, -1, 5]
    np.testing.assert_array_equal(o1.phi_direct(x, np.zeros_like(x))[0, :, 0, 0], [-0.0, .5, .5, .5, .5])
    o2 = _line_oracle(2)
    assert o2.wt.tolist() == [[0, 0], [1, 0], [.5, .5], [.5, .5], [.5, .5]]
    assert o2.time_list.tolist() == [[-1, -2], [0, -1], [1, 0], [2, 1], [3, 2]]
    assert o2.apply_op_Ldr(x)[0, :, 0, 0].tolist() == [0, 1, 1.5, 1.5, 1.5]
    assert o2.apply_op_Ldr_T(x)[0, :, 0, 0].tolist() == [-3.5, -1.5, -1.5, 1.5, 5]
    for s, o in ((1, o1), (2, o2)):
        np.testing.assert_allclose(o.apply_op_Ldr(x), g[f"line{s}_Ldr"], rtol=1e-15)
        np.testing.assert_allclose(o.apply_op_Ldr_T(x), g[f"line{s}_LdrT"], rtol=1e-15)
        np.testing.assert_array_equal(g[f"line{s}_d_ew"][:, :, 0], o.wt)
        np.testing.assert_array_equal(g[f"line{s}_time_list"], o.time_list)
    np.testing.assert_array_equal(o1.phi_direct(x, np.zeros_like(x)), g["line1_phi"])


def test_kat_initial_guess():
    g = load_golden("g6_kats.npz")
    x32 = orc.initial_guess(g["ig_y32"], 3, 6)
    assert x32.dtype == np.float32
    np.testing.assert_allclose(x32[0, :, 0, 0], [1, 3, 5, 7.000001, 9.000002, 11.000002], rtol=3e-7)
    np.testing.assert_allclose(x32, g["ig_x32"], rtol=3e-7)
    np.testing.assert_allclose(orc.initial_guess(g["ig_y32"].astype(np.float64), 3, 6), g["ig_x64"], rtol=1e-14)
    np.testing.assert_allclose(orc.initial_guess(g["ig_rand_y"], 12, 24), g["ig_rand_x"], rtol=1e-13)


def test_kat_initial_interpolation():
    g = load_golden("g6_kats.npz")
    x = orc.initial_interpolation(g["ii_y"], g["ii_mask"])
    np.testing.assert_allclose(x, g["ii_x"], rtol=1e-12, atol=1e-10)
    x32 = orc.initial_interpolation(g["ii_y"].astype(np.float32), g["ii_mask"])
    assert x32.dtype == np.float32
    np.testing.assert_allclose(x32, g["ii_x32"], rtol=2e-5, atol=2e-3)


# ---------------------------------------------------------------- G1 tables
@pytest.mark.parametrize("name", ["small", "pems", "ties", "road400"])
def test_g1_tables(name):
    g = load_golden(f"g1_tables_{name}.npz")
    n, k, sigma = int(g["n"]), int(g["k"]), float(g["sigma"])
    cl, dl = orc.k_nearest_neighbors(n, g["u_edges"], g["u_dist"], k)
    assert np.array_equal(cl, g["knn_cl"])
    np.testing.assert_allclose(dl, g["knn_dl"], rtol=1e-7)
    np.testing.assert_allclose(orc.undirected_weights(cl, g["knn_dl"], sigma), g["knn_u_ew"], rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(orc.directed_weights(cl, g["knn_dl"], sigma), g["knn_d_ew"], rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(orc.undirected_weights(cl, g["knn_dl"]), g["knn_u_ew_defsigma"], rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(orc.directed_weights(cl, g["knn_dl"]), g["knn_d_ew_defsigma"], rtol=2e-6, atol=1e-30)
    pcl, pdl = orc.connect_list(n, g["u_edges"], g["u_dist"])
    assert np.array_equal(pcl, g["phys_cl"]) and np.array_equal(pdl, g["phys_dl"])
    np.testing.assert_allclose(orc.undirected_weights(pcl, pdl, sigma), g["phys_u_ew"], rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(orc.directed_weights(pcl, pdl, sigma), g["phys_d_ew"], rtol=2e-6, atol=1e-30)


# ---------------------------------------------------------------- G2 dense operators
def _oracle_from_g2(g, mode, ablation="None"):
    info = admm_info_from(g)
    T, t_in = int(g["T"]), int(g["t_in"])
    if mode in ("line", "skip3"):
        return orc.OracleADMM(g["cl"], g["u_ew"], None, info, mode="line", ablation=ablation, t_in=t_in, T=T,
                              skip_connection=1 if mode == "line" else 3)
    return orc.OracleADMM(g["cl"], g["u_ew"], g["d_ew"], info, mode=mode, ablation=ablation, t_in=t_in, T=T)


@pytest.mark.parametrize("mode", ["knn", "physical", "line", "skip3"])
def test_g2_dense_operators(mode):
    g = load_golden(f"g2_ops_{mode}.npz")
    o = _oracle_from_g2(g, mode)
    for nm in ("Lu", "Ldr", "Ldr_T", "cLdr"):
        D = o.dense(getattr(o, "apply_op_" + nm))
        assert rel(D, g[nm]) < 1e-14, nm
    assert rel(o.dense(o.LHS_zu), g["LHS_zu"]) < 1e-14
    assert rel(o.dense(o.LHS_zd), g["LHS_zd"]) < 1e-14
    m = g["mask"]
    assert rel(o.dense(lambda e: o.LHS_x(e, mask=m)), g["LHS_x_mask_None"]) < 1e-14
    for abl in ("None", "DGLR", "DGTV"):
        oa = _oracle_from_g2(g, mode, abl)
        assert rel(oa.dense(oa.LHS_x), g["LHS_x_" + abl]) < 1e-14, abl


def test_q1_identity_on_t0_block():
    """Quirk Q1: the reference's kNN Ldr_T equals the exact transpose plus I on the t=0 block."""
    g = load_golden("g2_ops_knn.npz")
    N = int(g["n"])
    exact = g["Ldr"].T.copy()
    exact[:N, :N] += np.eye(N)
    assert rel(exact, g["Ldr_T"]) < 1e-14
    o = _oracle_from_g2(g, "knn")
    o.bug_compat = False
    assert rel(o.dense(o.apply_op_Ldr_T), g["Ldr"].T) < 1e-14
    gl = load_golden("g2_ops_skip3.npz")
    assert rel(gl["Ldr"].T, gl["Ldr_T"]) < 1e-14          # line modes are exact transposes


# ---------------------------------------------------------------- G3 CG
@pytest.mark.parametrize("mode", ["knn", "skip3"])
def test_g3_cg(mode):
    g = load_golden("g3_cg.npz")
    info = admm_info_from(g)
    T, t_in = int(g["T"]), int(g["t_in"])
    if mode == "knn":
        o = orc.OracleADMM(g["knn_cl"], g["knn_u_ew"], g["knn_d_ew"], info, mode="knn", t_in=t_in, T=T)
    else:
        o = orc.OracleADMM(g["skip3_cl"], g["skip3_u_ew"], None, info, mode="line", t_in=t_in, T=T,
                           skip_connection=3)
    rhs, x0, m = g[f"{mode}_rhs"], g[f"{mode}_x0"], g[f"{mode}_mask"]
    for nm, fn, kw in (("x", o.LHS_x, {}), ("xmask", o.LHS_x, {"mask": m}), ("zu", o.LHS_zu, {}),
                       ("zd", o.LHS_zd, {})):
        x, it, al, be = o.CG_solver(fn, rhs, x0, **kw)
        pre = f"{mode}_{nm}_f64_"
        assert int(it[0]) == int(g[pre + "iters"]), nm
        assert rel(x, g[pre + "x"]) < RTOL64
        # the reference returns torch.Tensor(alpha_list): float32-rounded (ADMM.py:362)
        np.testing.assert_allclose(al[:, 0], g[pre + "alpha"], rtol=2e-7)
        np.testing.assert_allclose(be[:, 0], g[pre + "beta"], rtol=2e-7, atol=1e-12)
        # float32 run of the restatement vs float32 run of the reference
        x32, it32, _, _ = o.CG_solver(fn, rhs.astype(np.float32), x0.astype(np.float32),
                                      **{k: v.astype(np.float32) for k, v in kw.items()})
        assert abs(int(it32[0]) - int(g[f"{mode}_{nm}_f32_iters"])) <= 1
        assert rel(x32, g[f"{mode}_{nm}_f32_x"]) < 5e-6
    x, it, _, _ = o.CG_solver(o.LHS_zu, rhs)
    assert int(it[0]) == int(g[f"{mode}_zu_zero_iters"]) and rel(x, g[f"{mode}_zu_zero_x"]) < RTOL64


# ---------------------------------------------------------------- G4 full solves
def make_oracle(meta, mode, ablation, bug_compat=True):
    info = admm_info_from(meta)
    T, t_in = int(meta["T"]), int(meta["t_in"])
    if mode == "knn":
        return orc.OracleADMM(meta["knn_cl"], meta["knn_u_ew"], meta["knn_d_ew"], info, mode="knn",
                              ablation=ablation, t_in=t_in, T=T, bug_compat=bug_compat)
    if mode == "physical":
        return orc.OracleADMM(meta["phys_cl"], meta["phys_u_ew"], meta["phys_d_ew"], info, mode="physical",
                              ablation=ablation, t_in=t_in, T=T, bug_compat=bug_compat)
    return orc.OracleADMM(meta["knn_cl"], meta["knn_u_ew"], None, info, mode="line", ablation=ablation,
                          t_in=t_in, T=T, skip_connection=1 if mode == "line" else 3)


def case_inputs(meta, task, tag):
    dt = np.float64 if tag == "f64" else np.float32
    t_in = int(meta["t_in"])
    if task == "pred":
        return meta["x_true"][:, :t_in].astype(dt), None
    mask = meta["mask"]                      # float32 like utils.py:129
    y = (meta["x_true"] * mask).astype(dt)
    return y, mask


def g4_cases(solves):
    keys = sorted({k.split("/")[0] for k in solves.files})
    return keys


def test_g4_full_solves_f64(g4_meta, g4_solves):
    checked = 0
    for key in g4_cases(g4_solves):
        mode, abl, task, tag, iters = key.split("-")
        if tag != "f64":
            continue
        o = make_oracle(g4_meta, mode, abl)
        y, mask = case_inputs(g4_meta, task, tag)
        o.max_ADMM_iter = int(iters)
        x = o.combined_loop(y, mask=mask)
        G = lambda f: g4_solves[f"{key}/{f}"]
        h = o.hist
        assert rel(x, G("x")) < 1e-11, key
        assert np.array_equal(np.array([int(v[0]) for v in h.CG_iter_x]), G("CG_iter_x")), key
        assert np.array_equal(np.array([int(v[0]) for v in h.CG_iter_zu]), G("CG_iter_zu")), key
        if abl != "DGLR":
            assert np.array_equal(np.array([int(v[0]) for v in h.CG_iter_zd]), G("CG_iter_zd")), key
            np.testing.assert_allclose(h.DGLR_list, G("DGLR"), rtol=1e-9)
            assert rel(o.state["zd"], G("zd")) < 1e-11
        np.testing.assert_allclose(np.array(h.p_res_list), G("p_res"), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(np.array(h.d_res_list), G("d_res"), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(h.x_shift_list, G("x_shift"), rtol=1e-8)
        np.testing.assert_allclose(np.array(h.delta_x_per_step), G("dxps"), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(h.GLR_list, G("GLR"), rtol=1e-9)
        np.testing.assert_allclose(h.recover_list, G("recover"), rtol=1e-8, atol=1e-12)
        assert rel(o.state["zu"], G("zu")) < 1e-11
        if abl in ("None", "DGLR"):
            np.testing.assert_allclose(h.DGTV_list, G("DGTV"), rtol=1e-9)
            assert rel(o.state["phi"], G("phi")) < 1e-10
        a0 = np.array([a[:, 0] for a in h.alpha_x][0])
        np.testing.assert_allclose(a0, G("alpha_x")[0][: len(a0)], rtol=2e-7)
        checked += 1
    assert checked >= 30


def test_g4_full_solves_f32(g4_meta, g4_solves):
    """float32 restatement vs the reference's own float32 runs: the reference's fp32-vs-fp64 drift
     sets the tolerance."""
    checked = 0
    for key in g4_cases(g4_solves):
        mode, abl, task, tag, iters = key.split("-")
        if tag != "f32":
            continue
        o = make_oracle(g4_meta, mode, abl)
        y, mask = case_inputs(g4_meta, task, tag)
        if mask is not None:
            mask = mask.astype(np.float32)
        o.max_ADMM_iter = int(iters)
        x = o.combined_loop(y, mask=mask)
        assert x.dtype == np.float32
        G = lambda f: g4_solves[f"{key}/{f}"]
        assert rel(x, G("x")) < 1e-5, key
        np.testing.assert_allclose(np.array(o.hist.p_res_list), G("p_res"), rtol=2e-3, atol=1e-4)
        d = np.abs(np.array([int(v[0]) for v in o.hist.CG_iter_zu]) - G("CG_iter_zu"))
        assert d.max() <= 2, key
        checked += 1
    assert checked >= 30


def test_g5_batched_equals_looped():
    """Batched semantics = B independent B=1 reference runs (quirk Q6)."""
    g = load_golden("g5_batched.npz")
    meta = load_golden("g4_meta.npz")
    o = make_oracle(meta, "knn", "None")
    o.max_ADMM_iter = int(g["iters"])
    x = o.combined_loop(g["y"])
    assert rel(x, g["x"]) < 1e-11
    itx = np.array(o.hist.CG_iter_x)               # (iters, B)
    assert np.array_equal(itx.T, g["CG_iter_x"])
    assert np.array_equal(np.array(o.hist.CG_iter_zd).T, g["CG_iter_zd"])
    # whole-batch Frobenius norms == sqrt(sum of per-sample squares)
    pr = np.sqrt((g["p_res"] ** 2).sum(0))
    np.testing.assert_allclose(np.array(o.hist.p_res_list), pr, rtol=1e-8)
    np.testing.assert_allclose(o.hist.GLR_list, g["GLR"].mean(0), rtol=1e-9)
    assert rel(o.state["phi"], g["phi"].reshape(o.state["phi"].shape)) < 1e-10


# ------------------------------------------------------------------ round-2 fixture groups (g8, g9, g10)
def _oracle_from(g, prefix, mode, **kw):
    info = {k: float(g[k]) for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2")} if "rho" in g.files else \
        dict(rho=1.0, rho_u=1.0, rho_d=1.0, mu_u=1.0, mu_d1=1.0, mu_d2=1.0)
    cl = g[prefix + "cl"]
    d_ew = g[prefix + "d_ew"] if (prefix + "d_ew") in g.files else None
    return orc.OracleADMM(cl, g[prefix + "u_ew"], d_ew, info, mode=mode, T=int(g["T"]), t_in=int(g["t_in"]), **kw)


@pytest.mark.parametrize("mode", ["knn", "physical", "line"])
def test_apply_op_Ln_matches_reference(mode):
    """g8: apply_op_Ln (ADMM.py:248-288) as a dense matrix and on a multi-channel random tensor."""
    g = load_golden("g8_ln.npz")
    o = _oracle_from(g, mode + "_", mode)
    assert rel(o.dense(o.apply_op_Ln), g[mode + "_Ln"]) < 1e-15
    x = g[mode + "_x"]
    got = np.stack([o.apply_op_Ln(x[..., c:c + 1]) for c in range(x.shape[-1])], -1)[..., 0, :]
    assert rel(got, g[mode + "_y"]) < 1e-15


@pytest.mark.parametrize("mode,abl", [("knn", "None"), ("knn", "DGLR"), ("knn", "DGTV"), ("line", "None"), ("line", "DGLR"), ("line", "DGTV")])
def test_two_loops_matches_reference(mode, abl):
    """g9: two_loops (ADMM.py:410-508), 3 outer x 4 inner iterations: x / gamma / phi at every outer iteration as the
    reference hands them to phi_direct, and every CG iteration count."""
    g = load_golden("g9_two_loops.npz")
    info = {k: float(g[k]) for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2")}
    o = orc.OracleADMM(g["cl"], g["u_ew"], g["d_ew"] if mode == "knn" else None, info, mode=mode, ablation=abl,
                       T=int(g["T"]), t_in=int(g["t_in"]))
    o.max_ADMM_iter, o.max_inner_iter = int(g["max_outer"]), int(g["max_inner"])
    o.two_loops(g["y"])
    tag = f"{mode}_{abl}"
    if abl != "DGTV":
        for kq in ("x", "gamma", "phi"):
            got = np.stack([r[kq] for r in o.outer])
            assert rel(got, g[f"{tag}_{kq}"]) < 1e-11, kq
    assert np.array_equal(np.array(o.hist.CG_iter_x).reshape(-1), g[f"{tag}_cg_x"])
    assert np.array_equal(np.array(o.hist.CG_iter_zu).reshape(-1), g[f"{tag}_cg_zu"])
    if abl != "DGLR":
        assert np.array_equal(np.array(o.hist.CG_iter_zd).reshape(-1), g[f"{tag}_cg_zd"])


@pytest.mark.parametrize("nm", ["x", "zu", "zd"])
def test_cg_batch_max_stop_matches_reference(nm):
    """g10: the reference's batch-global CG stop (ADMM.py:360) on 8 samples that converge at different iterations."""
    g = load_golden("g10_cg_batchmax.npz")
    o = _oracle_from(g, "", "knn", cg_convergence="batch_max")
    fn = {"x": o.LHS_x, "zu": o.LHS_zu, "zd": o.LHS_zd}[nm]
    x, it, al, be = o.CG_solver(fn, g["rhs"], g["x0"])
    assert (it == int(g[nm + "_iters"])).all()
    assert rel(x, g[nm + "_x"]) < 1e-12
    np.testing.assert_allclose(al, g[nm + "_alpha"], rtol=1e-9)
    np.testing.assert_allclose(be, g[nm + "_beta"], rtol=1e-9)
    o2 = _oracle_from(g, "", "knn")
    _, it2, _, _ = o2.CG_solver(fn, g["rhs"], g["x0"])
    assert np.array_equal(it2, g[nm + "_iters_per_sample"])
