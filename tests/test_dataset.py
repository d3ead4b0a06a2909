"""TrafficDataset front end (mgadmm.dataset, csrc/frontend.hip) against the fixture produced by
running the reference's TrafficDataset on synthetic files (g7_dataset.npz; the files are re-created here from
the stored arrays).  Statistics and transforms are float64: tolerance 1e-12 relative (summation order differs
from torch's CPU reduction); windows and masks are exact."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden


def write_files(g, td):
    pd.DataFrame({"from": g["csv_from"], "to": g["csv_to"], "cost": g["csv_cost"]}).to_csv(os.path.join(td, "dist.csv"), index=False)
    idx = {int(v): i for i, v in enumerate(g["ids"])}
    pd.DataFrame({"from": [idx[int(v)] for v in g["csv_from"]], "to": [idx[int(v)] for v in g["csv_to"]],
                  "cost": g["csv_cost"]}).to_csv(os.path.join(td, "dist_plain.csv"), index=False)
    np.savetxt(os.path.join(td, "ids.txt"), g["ids"], fmt="%d")
    np.savez(os.path.join(td, "series.npz"), data=g["data"])


def test_interpolation_mask_is_the_reference_mask():
    from mgadmm.dataset import TrafficDataset
    g = load_golden("g7_dataset.npz")
    m = TrafficDataset.interpolation_mask((24, 12, 1), 0.4)
    assert m.dtype == torch.float32
    np.testing.assert_array_equal(m.numpy(), g["None/int_mask"])
    np.testing.assert_array_equal(m.numpy(), g["normalize/int_mask"])      # reseeded on every call: one mask for all windows


@pytest.mark.gpu
@pytest.mark.parametrize("transform", [None, "standardize", "normalize"])
def test_dataset_matches_reference(tmp_path, transform):
    from mgadmm.dataset import TrafficDataset
    g = load_golden("g7_dataset.npz")
    write_files(g, str(tmp_path))
    ds = TrafficDataset(str(tmp_path), "series.npz", "dist.csv", id_file="ids.txt", transform=transform, verbose=False)
    tag = str(transform)
    gi = ds.graph_info
    assert gi["n_nodes"] == int(g["gi_n_nodes"]) and gi["n_edges"] == int(g["gi_n_edges"])
    np.testing.assert_array_equal(gi["u_edges"].numpy(), g["gi_u_edges"])
    np.testing.assert_array_equal(gi["u_dist"].numpy(), g["gi_u_dist"])
    assert ds.data.is_cuda and ds.data.dtype == torch.float64 and tuple(ds.data.shape) == (80, 12, 1)
    np.testing.assert_allclose(ds.data.cpu().numpy(), g[f"{tag}/data"], rtol=1e-12, atol=1e-13)
    if transform == "standardize":
        np.testing.assert_allclose(ds.data_mean.cpu().numpy(), g[f"{tag}/mean"], rtol=1e-13)
        np.testing.assert_allclose(ds.data_std.cpu().numpy(), g[f"{tag}/std"], rtol=1e-12)
    if transform == "normalize":
        np.testing.assert_array_equal(ds.data_max.cpu().numpy(), g[f"{tag}/max"])
        np.testing.assert_array_equal(ds.data_min.cpu().numpy(), g[f"{tag}/min"])
    x, y = ds.get_predict_data(5)
    np.testing.assert_allclose(x.cpu().numpy(), g[f"{tag}/pred_x"], rtol=1e-12, atol=1e-13)
    assert tuple(y.shape) == (12, 12, 1) and torch.equal(y, x[:12])
    ix, iy, im = ds.get_interpolated_data(7, 0.4)
    np.testing.assert_array_equal(im.cpu().numpy(), g[f"{tag}/int_mask"])
    np.testing.assert_allclose(iy.cpu().numpy(), g[f"{tag}/int_y"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ds.recover_data(x).cpu().numpy(), g[f"{tag}/recovered"], rtol=1e-12)
    np.testing.assert_allclose(ds.recover_data(x).cpu().numpy(), g["None/pred_x"], rtol=1e-12)        # round trip
    # batches of windows: every row equals the single-window accessor; ragged / repeated / last start indices
    starts = [5, 0, 56, 7, 7]
    xb, yb = ds.get_predict_batch(starts)
    assert tuple(xb.shape) == (5, 24, 12, 1) and tuple(yb.shape) == (5, 12, 12, 1)
    for b, s in enumerate(starts):
        assert torch.equal(xb[b], ds.data[s:s + 24]) and torch.equal(yb[b], ds.data[s:s + 12])
    xi, yi, mi = ds.get_interpolated_batch(starts, 0.4)
    assert torch.equal(xi, xb) and torch.equal(mi[3], im) and torch.equal(yi[3], iy) and torch.equal(yi, xi * mi)


@pytest.mark.gpu
def test_dataset_without_id_file_and_errors(tmp_path):
    from mgadmm import _lib
    from mgadmm.dataset import TrafficDataset, gather_windows, series_stats
    g = load_golden("g7_dataset.npz")
    write_files(g, str(tmp_path))
    ds = TrafficDataset(str(tmp_path), "series.npz", "dist_plain.csv", verbose=False)
    assert ds.graph_info["n_nodes"] == int(g["plain_n_nodes"])
    np.testing.assert_array_equal(ds.graph_info["u_edges"].numpy(), g["plain_u_edges"])
    assert ds.recover_data(ds.data[:3]) is ds.data[:3] or torch.equal(ds.recover_data(ds.data[:3]), ds.data[:3])
    with pytest.raises(_lib.MgadmmError, match="start index"):
        gather_windows(ds.data, [57], 24)
    with pytest.raises(_lib.MgadmmError, match="start index"):
        gather_windows(ds.data, [-1], 24)
    with pytest.raises(ValueError):
        series_stats(ds.data.cpu())
    # float32 series, wide and long enough for several row chunks and column blocks; statistics vs torch
    gen = torch.Generator().manual_seed(0)
    s = (torch.randn(1500, 333, generator=gen) * 40 + 200).cuda()
    mn, mx, mean, std = series_stats(s)
    assert torch.equal(mn[0], s.min(0)[0]) and torch.equal(mx[0], s.max(0)[0])
    np.testing.assert_allclose(mean[0].cpu().numpy(), s.double().mean(0).cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(std[0].cpu().numpy(), s.double().std(0).cpu().numpy(), rtol=1e-6)
    mn2, mx2, mean2, std2 = series_stats(s)
    assert torch.equal(mean, mean2) and torch.equal(std, std2)          # fixed-order reductions: bitwise repeatable
    w = gather_windows(s, torch.arange(0, 1400, 7), 24)
    assert torch.equal(w, s.unfold(0, 24, 1).permute(0, 2, 1)[::7][:200])


@pytest.mark.gpu
def test_windows_feed_the_solver(tmp_path):
    """End to end: files -> TrafficDataset on the GPU -> batch of windows -> combined_loop; equals solving the
    windows one at a time (the reference's B = 1 usage)."""
    import mgadmm
    g = load_golden("g7_dataset.npz")
    write_files(g, str(tmp_path))
    ds = mgadmm.TrafficDataset(str(tmp_path), "series.npz", "dist.csv", id_file="ids.txt", verbose=False)
    n = ds.graph_info["n_nodes"]
    r = (n / 24) ** 0.5
    info = {"rho": 2 * r, "rho_u": 3 * r, "rho_d": 2 * r, "mu_u": 1, "mu_d1": 2, "mu_d2": 1}
    blk = mgadmm.ADMM_algorithm(ds.graph_info, info, use_kNN=True, k=3, u_sigma=50, d_sigma=50)
    blk.max_ADMM_iter = 8
    starts = [0, 9, 30, 56]
    x_true, y = ds.get_predict_batch(starts)
    xb = blk.combined_loop(y, print_info=False)
    assert xb.is_cuda and tuple(xb.shape) == (4, 24, n, 1)
    for b, s in enumerate(starts):
        blk._reset_history()
        x1 = blk.combined_loop(ds.get_predict_data(s)[1].unsqueeze(0), print_info=False)
        assert float((xb[b] - x1[0]).norm() / x1[0].norm()) < 1e-6
    blk.close()


@pytest.mark.gpu
def test_example_workflow_runs(monkeypatch):
    """examples/pems_workflow.py: files -> GPU-resident data set -> kNN graph -> batched prediction and interpolation."""
    import importlib.util
    import sys as _sys
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("pems_workflow", os.path.join(ROOT, "examples", "pems_workflow.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(_sys, "argv", ["pems_workflow.py", "--nodes", "60", "--steps", "200", "--batch", "8", "--iters", "5"])
    mae_pred, mae_int = mod.main()
    assert np.isfinite(mae_pred) and np.isfinite(mae_int)
