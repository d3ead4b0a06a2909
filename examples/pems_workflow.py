#!/usr/bin/env python3
"""The reference's notebook / experiment flow (example-PEMS04.ipynb, experiment_0319.py) on the MI355X path:

    files -> TrafficDataset (series resident on the GPU) -> kNN graph -> ADMM_algorithm
          -> prediction of the next 12 steps for a whole batch of sliding windows
          -> interpolation of 40 % masked entries

The PEMS files are not redistributable, so the script writes a synthetic PEMS-shaped data set (distance csv,
sensor-id file, (T, N, 3) series) into a temporary directory first; point --data-folder at a real
PEMS0X folder (PEMS04.npz / PEMS04.csv) to run on the real thing.

    python examples/pems_workflow.py [--nodes 307 --steps 2000 --batch 256 --iters 50]
"""
import argparse
import math
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))

from mgadmm.ADMM import ADMM_algorithm            # noqa: E402   (was: from ADMM import *)
from mgadmm.dataset import TrafficDataset          # noqa: E402   (was: from utils import *)
from mgadmm.gpu_graph import k_nearest_neighbors   # noqa: E402


def write_synthetic_pems(folder, n, steps, seed=0):
    import pandas as pd
    rng = np.random.default_rng(seed)
    ids = 1000 + 3 * np.arange(n)
    e = [(i, i + 1) for i in range(n - 1)]
    have = set(e)
    while len(e) < n - 1 + n // 9:
        a, b = int(rng.integers(n)), int(rng.integers(n))
        if a != b and (a, b) not in have and (b, a) not in have:
            e.append((a, b)); have.add((a, b))
    e = np.array(e)
    pd.DataFrame({"from": ids[e[:, 0]], "to": ids[e[:, 1]], "cost": rng.uniform(3.0, 2900.0, len(e))}).to_csv(
        os.path.join(folder, "PEMS.csv"), index=False)
    np.savetxt(os.path.join(folder, "PEMS.txt"), ids, fmt="%d")
    t = np.arange(steps)[:, None]
    # neighbouring sensors see similar, slightly delayed traffic: level and phase vary smoothly along the path
    base = np.clip(220 + np.cumsum(rng.normal(0, 6, n)), 60, 400)[None, :]
    phase = np.cumsum(rng.normal(0, 0.15, n))[None, :]
    flow = base * (1 + 0.35 * np.sin(2 * np.pi * t / 288 + phase)) + 8 * rng.standard_normal((steps, n))
    data = np.stack([flow, rng.random((steps, n)), rng.random((steps, n))], -1)
    np.savez(os.path.join(folder, "PEMS.npz"), data=data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-folder", default=None)
    ap.add_argument("--nodes", type=int, default=307)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--transform", default="none", choices=["none", "standardize", "normalize"],
                    help="TrafficDataset transform; the notebooks' ADMM weights are tuned for the raw scale")
    args = ap.parse_args()
    tmp = None
    if args.data_folder is None:
        tmp = tempfile.TemporaryDirectory()
        write_synthetic_pems(tmp.name, args.nodes, args.steps)
        folder, data_file, graph_csv, id_file = tmp.name, "PEMS.npz", "PEMS.csv", "PEMS.txt"
    else:
        name = os.path.basename(os.path.normpath(args.data_folder))
        folder, data_file, graph_csv, id_file = args.data_folder, name + ".npz", name + ".csv", None

    ds = TrafficDataset(folder, data_file, graph_csv, id_file=id_file, transform=None if args.transform == "none" else args.transform)
    n = ds.graph_info["n_nodes"]
    print(f"data shape: {tuple(ds.data.shape)} on {ds.data.device}, node number: {n}, edge number: {ds.graph_info['n_edges']}")
    nn, nd = k_nearest_neighbors(n, ds.graph_info["u_edges"], ds.graph_info["u_dist"], 4)
    print(f"nearest nodes: {tuple(nn.shape)}, nearest_dists: {tuple(nd.shape)}")

    r = math.sqrt(n / 24)
    admm_info = {"rho": 2 * r, "rho_u": 3 * r, "rho_d": 2 * r, "mu_u": 1, "mu_d1": 2, "mu_d2": 1}
    blk = ADMM_algorithm(ds.graph_info, admm_info, use_kNN=True, k=4, u_sigma=50, d_sigma=50)
    blk.max_ADMM_iter = args.iters

    starts = torch.arange(0, args.batch) * ((ds.data.shape[0] - 24) // args.batch)
    x_true, y = ds.get_predict_batch(starts)                      # (B, 24, N, 1), (B, 12, N, 1) on the GPU
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x_hat = blk.combined_loop(y, print_info=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    its = len(blk.p_res_list)
    err = (ds.recover_data(x_hat[:, 12:]) - ds.recover_data(x_true[:, 12:])).abs()
    print(f"prediction: {args.batch} windows x {its} ADMM iterations in {dt * 1e3:.1f} ms ({args.batch * its / dt:,.0f} sample-iterations/s); "
          f"MAE of the 12 predicted steps {err.mean().item():.2f}, fit of the 12 observed steps ||Hx - y|| = {blk.recover_list[-1]:.3g}")
    print(f"  CG iterations x/zu/zd of the last ADMM iteration: {blk.CG_iter_x[-1].float().mean():.1f} / "
          f"{blk.CG_iter_zu[-1].float().mean():.1f} / {blk.CG_iter_zd[-1].float().mean():.1f}; "
          f"primal residuals {['%.3g' % v for v in blk.p_res_list[-1]]}")

    ix, iy, mask = ds.get_interpolated_batch(starts, 0.4)
    blk._reset_history()
    x_int = blk.combined_loop(iy, mask=mask, print_info=False)
    hidden = mask == 0
    e_int = (ds.recover_data(x_int) - ds.recover_data(ix)).abs()[hidden]
    print(f"interpolation (40 % masked): MAE on the hidden entries {e_int.mean().item():.2f}, "
          f"GLR / DGTV / DGLR = {blk.GLR_list[-1].item():.4g} / {blk.DGTV_list[-1].item():.4g} / {blk.DGLR_list[-1].item():.4g}")
    blk.close()
    if tmp is not None:
        tmp.cleanup()
    return err.mean().item(), e_int.mean().item()


if __name__ == "__main__":
    main()
