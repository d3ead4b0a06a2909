"""GPU parity of the SURVEY.md 8(f) rank-4 rows and the section-5 ABI completions, against golden vectors produced by the
imported reference (tests/golden/gen_golden.py groups g8, g9, g10) and against the solver itself (resume):

  apply_op_Ln           ADMM.py:248-288   dense matrices, kNN / physical / line graph            (g8)
  two_loops             ADMM.py:410-508   x / phi per outer iteration + every CG count           (g9)
  cg_convergence        ADMM.py:360       the reference's batch-global CG stop on 8 samples      (g10)
  warm start / resume   SURVEY 5          k1 + k2 iterations in two calls == k1 + k2 in one, bit for bit
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import rel

pytestmark = pytest.mark.gpu


def T_(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _product(g, prefix, mode, ablation="None", compute_dtype=torch.float64, info=None, **kw):
    from mgadmm.ADMM import ADMM_algorithm
    if info is None:
        info = {k: float(g[k]) for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2")} if "rho" in g.files else \
            dict(rho=1.0, rho_u=1.0, rho_d=1.0, mu_u=1.0, mu_d1=1.0, mu_d2=1.0)
    cl = torch.from_numpy(g[prefix + "cl"])
    T, t_in, n = int(g["T"]), int(g["t_in"]), int(g["n"])
    blk = ADMM_algorithm({"n_nodes": n}, info, use_kNN=(mode != "physical"), k=cl.shape[1] - 1, u_sigma=1.0, d_sigma=1.0,
                         ablation=ablation, t_in=t_in, T=T, use_line_graph=(mode == "line"), tables=(cl, torch.ones(cl.shape)),
                         compute_dtype=compute_dtype, **kw)
    blk.u_ew = torch.from_numpy(np.asarray(g[prefix + "u_ew"], dtype=np.float32)).unsqueeze(0).repeat(T, 1, 1)
    if mode != "line":
        blk.d_ew = torch.from_numpy(np.asarray(g[prefix + "d_ew"], dtype=np.float32)).unsqueeze(0).repeat(T - 1, 1, 1)
    return blk


# ---------------------------------------------------------------------------------------------- apply_op_Ln (g8)
@pytest.mark.parametrize("mode", ["knn", "physical", "line"])
@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-13), (torch.float32, 2e-6)])
def test_apply_op_Ln_matches_reference(mode, dt, tol):
    g = load_golden("g8_ln.npz")
    T, n = int(g["T"]), int(g["n"])
    for reorder in (False, "cluster"):
        blk = _product(g, mode + "_", mode, compute_dtype=dt, reorder=reorder)
        size = T * n
        eye = torch.eye(size, dtype=dt).reshape(size, T, n, 1)
        D = blk.apply_op_Ln(eye).reshape(size, size).T.double().numpy()
        assert rel(D, g[mode + "_Ln"]) < tol, (mode, reorder)
        y = blk.apply_op_Ln(T_(g[mode + "_x"], dt))                      # (3, T, n, 2): channels folded into nodes
        assert y.shape == g[mode + "_y"].shape and rel(y, g[mode + "_y"]) < tol
        blk.close()


# ---------------------------------------------------------------------------------------------- two_loops (g9)
@pytest.mark.parametrize("mode,abl", [("knn", "None"), ("knn", "DGLR"), ("knn", "DGTV"), ("line", "None"), ("line", "DGLR")])
@pytest.mark.parametrize("dt,xtol,slack", [(torch.float64, 1e-10, 0), (torch.float32, 1e-5, 1)])
def test_two_loops_matches_reference(mode, abl, dt, xtol, slack):
    g = load_golden("g9_two_loops.npz")
    n_outer, n_inner = int(g["max_outer"]), int(g["max_inner"])
    tag = f"{mode}_{abl}"
    y = T_(g["y"])
    for k_out in range(1, n_outer + 1):                 # the state after k outer iterations = the reference's k-th capture
        blk = _product(g, "", mode, ablation=abl, compute_dtype=dt)
        blk.max_ADMM_iter, blk.max_inner_iter = k_out, n_inner
        assert blk.two_loops(y) is None                 # the reference returns None (ADMM.py:410-508)
        if abl != "DGTV":
            assert rel(blk.state["x"], g[f"{tag}_x"][k_out - 1]) < xtol, (tag, k_out)
            assert rel(blk.state["phi"], g[f"{tag}_phi"][k_out - 1]) < 50 * xtol, (tag, k_out)
        for nm in ("x", "zu") + (("zd",) if abl != "DGLR" else ()):
            got = np.array(getattr(blk, "CG_iter_" + nm))
            assert got.shape == (k_out * n_inner,)
            # +-1, and +-2 on the 2-iteration diagonal solves (LHS_x of 'DGTV' is diagonal) in float32
            ok = slack + (1 if (slack and abl == "DGTV" and nm == "x") else 0)
            assert np.abs(got - g[f"{tag}_cg_{nm}"][: k_out * n_inner]).max() <= ok, (tag, nm)
        blk.close()


# ---------------------------------------------------------------------------------------------- batch-global CG stop (g10)
@pytest.mark.parametrize("nm", ["x", "zu", "zd"])
def test_cg_batch_max_stop_matches_reference(nm):
    g = load_golden("g10_cg_batchmax.npz")
    rhs, x0 = T_(g["rhs"]), T_(g["x0"])
    kstar = int(g[nm + "_iters"])
    blk = _product(g, "", "knn", cg_convergence="batch_max")
    fn = {"x": blk.LHS_x, "zu": blk.LHS_zu, "zd": blk.LHS_zd}[nm]
    x, it, al, be = blk.CG_solver(fn, rhs, x0)
    assert (it == kstar).all()                                   # ONE count for the whole batch (ADMM.py:360)
    assert rel(x, g[nm + "_x"]) < 1e-10
    np.testing.assert_allclose(al.numpy()[:kstar], g[nm + "_alpha"], rtol=1e-8)     # every sample iterates to the end
    np.testing.assert_allclose(be.numpy()[:kstar], g[nm + "_beta"], rtol=1e-8)
    blk.close()
    # default semantics on the same systems: every sample stops on its own residual
    blk = _product(g, "", "knn")
    fn = {"x": blk.LHS_x, "zu": blk.LHS_zu, "zd": blk.LHS_zd}[nm]
    _, it2, al2, _ = blk.CG_solver(fn, rhs, x0)
    assert np.array_equal(it2.numpy(), g[nm + "_iters_per_sample"])
    assert np.isnan(al2.numpy()[int(it2.min()):, int(it2.argmin())]).all()
    blk.close()
    # float32 kernels: same stop rule, float32 tolerances
    blk = _product(g, "", "knn", compute_dtype=torch.float32, cg_convergence="batch_max", path="stream")
    fn = {"x": blk.LHS_x, "zu": blk.LHS_zu, "zd": blk.LHS_zd}[nm]
    x32, it32, _, _ = blk.CG_solver(fn, rhs, x0)
    assert len(set(it32.tolist())) == 1 and abs(int(it32[0]) - kstar) <= 1 and rel(x32, g[nm + "_x"]) < 1e-5
    blk.close()


def test_batch_max_full_solve_runs_on_the_streaming_path_and_couples_the_batch():
    """A whole solve under 'batch_max': the LDS-resident kernel (one workgroup per sample) cannot express a batch-global
    stop, so the streaming path is taken; every CG count of an ADMM iteration is the same for all samples and is the
    maximum of the per-sample counts."""
    from mgadmm import _lib
    g = load_golden("g4_meta.npz")
    gb = load_golden("g5_batched.npz")
    import helpers
    y = T_(gb["y"], torch.float32)
    per = helpers.make_product(g, "knn")
    per.max_ADMM_iter = 3
    per.check_stop = False
    per.combined_loop(y, print_info=False)
    bm = helpers.make_product(g, "knn", cg_convergence="batch_max")
    bm.max_ADMM_iter = 3
    bm.check_stop = False
    xb = bm.combined_loop(y, print_info=False)
    h = bm._solvers[(1, torch.float32)][0]
    assert _lib.lib.mgadmm_solver_path(h, y.shape[0]) == _lib.PATH_STREAM
    assert _lib.lib.mgadmm_solver_path(per._solvers[(1, torch.float32)][0], y.shape[0]) == _lib.PATH_LDS
    assert torch.isfinite(xb).all()
    for nm in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd"):
        first_b, first_p = getattr(bm, nm)[0], getattr(per, nm)[0]
        assert len(set(first_b.tolist())) == 1
        assert abs(int(first_b[0]) - int(first_p.max())) <= 1          # first iteration: same start, batch stops with its slowest sample
    per.close(); bm.close()


# ---------------------------------------------------------------------------------------------- warm start / resume
@pytest.mark.parametrize("path,abl,task", [("lds", "None", "pred"), ("stream", "None", "pred"), ("stream", "DGLR", "mask"),
                                           ("lds", "DGTV", "mask"), ("stream", "UT", "pred")])
def test_resume_equals_uninterrupted_solve(path, abl, task):
    import helpers
    g = load_golden("g4_meta.npz")
    y, mask = helpers.case_inputs(g, task, np.float32)
    yt = torch.from_numpy(y)
    mt = torch.from_numpy(mask) if mask is not None else None
    blk = helpers.make_product(g, "knn", ablation=abl, path=path)
    blk.check_stop = False
    blk.max_ADMM_iter = 7
    x_full = blk.solve(yt, mask=mt)[0]
    full = (np.array(blk.p_res_list), np.array(blk.d_res_list), np.array(blk.x_shift_list), list(blk.CG_iter_x))
    blk._reset_history()
    blk.max_ADMM_iter = 4
    blk.solve(yt, mask=mt)
    saved = {k: v.clone() for k, v in blk.state.items()}           # the checkpoint
    first = (np.array(blk.p_res_list), np.array(blk.d_res_list), np.array(blk.x_shift_list))
    blk._reset_history()
    blk.max_ADMM_iter = 3
    x_res = blk.solve(yt, mask=mt, warm_start=saved)[0]
    assert torch.equal(x_res, x_full)                               # bit for bit
    np.testing.assert_array_equal(np.concatenate([first[0], np.array(blk.p_res_list)]), full[0])
    np.testing.assert_array_equal(np.concatenate([first[1], np.array(blk.d_res_list)]), full[1])
    np.testing.assert_array_equal(np.concatenate([first[2], np.array(blk.x_shift_list)]), full[2])
    got = blk.CG_iter_x
    assert all(torch.equal(torch.as_tensor(a), torch.as_tensor(b)) for a, b in zip(got, full[3][4:]))
    with pytest.raises(ValueError, match="warm_start misses"):
        blk.solve(yt, mask=mt, warm_start={"x": saved["x"]})
    blk.close()


def test_resume_on_the_fused_streaming_path_with_the_folded_vector_update():
    """The same resume property at B = 256 with the cluster order: the x / zd solves run through k_cldr with the CG vector
    update folded into its loads (p alternates between two buffers, the last x update is applied after the loop)."""
    import helpers
    g = load_golden("g4_meta.npz")
    rng = np.random.default_rng(11)
    yt = torch.from_numpy((100 + 50 * rng.random((256, 12, 30, 1))).astype(np.float32))
    blk = helpers.make_product(g, "knn", path="stream", reorder="cluster")
    blk.check_stop = False
    blk.max_ADMM_iter = 5
    x_full = blk.solve(yt)[0]
    full = np.array(blk.p_res_list)
    blk._reset_history()
    blk.max_ADMM_iter = 2
    blk.solve(yt)
    saved = {k: v.clone() for k, v in blk.state.items()}
    first = np.array(blk.p_res_list)
    blk._reset_history()
    blk.max_ADMM_iter = 3
    x_res = blk.solve(yt, warm_start=saved)[0]
    assert torch.equal(x_res, x_full)
    np.testing.assert_array_equal(np.concatenate([first, np.array(blk.p_res_list)]), full)
    blk.close()


@pytest.mark.parametrize("path", ["lds", "stream"])
def test_global_stop_driver_equals_the_solvers_own_stop_test(path):
    """mgadmm.dist.sharded_solve_global_stop (the reference's stop test on whole-batch norms for sharded runs: one iteration
    per solver call, resumed) on ONE rank against the solver's own loop with check_stop=True: the same iteration count, the
    same x bit for bit, the same residual history."""
    import helpers
    from mgadmm.dist import sharded_solve_global_stop
    g = load_golden("g4_meta.npz")
    y, _ = helpers.case_inputs(g, "pred", np.float32)
    yt = torch.from_numpy(y)
    blk = helpers.make_product(g, "knn", path=path)
    blk.ADMM_tol = 60.0
    blk.check_stop = True
    blk.max_ADMM_iter = 40
    x_ref = blk.solve(yt)[0]
    p_ref, d_ref = np.array(blk.p_res_list), np.array(blk.d_res_list)
    n_ref = len(p_ref)
    assert 1 < n_ref < 40
    blk._reset_history()
    x, n_it = sharded_solve_global_stop(blk, yt)
    assert n_it == n_ref and torch.equal(x, x_ref)
    np.testing.assert_array_equal(np.array(blk.p_res_list), p_ref)
    np.testing.assert_array_equal(np.array(blk.d_res_list), d_ref)
    assert (blk.max_ADMM_iter, blk.check_stop) == (40, True)
    blk.close()
