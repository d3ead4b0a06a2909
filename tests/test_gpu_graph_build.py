"""Graph construction on the GPU (csrc/build.hip through the C ABI): kNN tables and weight
tables against the golden tables produced by the reference (G1: distinct distances, integer distances with
ties, a 400-node road-like graph), against the host builders on larger graphs, and the error conventions.

Integer / index work is bit-exact (neighbour ids, float32 distances).  Weight tables are float32
`exp(-d/sigma)` + normalisation: device `expf` vs torch's CPU `exp` differ by <= 1 ulp, tolerance 2e-6
relative (the same tolerance the oracle has against the reference)."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
W_RTOL = 2e-6


def assert_weights_close(actual, desired, dist, sigma):
    """rtol 2e-6 wherever the raw weight exp(-d/sigma) is a normal float32; raw weights in the denormal range
    (d/sigma > 87.3, below 1.2e-38 before normalisation) only have a few significant bits in either
    implementation: there the two may differ by a factor of two, and both are ~0 for every purpose."""
    actual, desired = np.asarray(actual), np.asarray(desired)
    raw = np.exp(-(np.asarray(dist, dtype=np.float32) / np.float32(sigma)), dtype=np.float32)
    normal = ~((raw > 0) & (raw < 1.2e-38))
    np.testing.assert_allclose(actual[normal], desired[normal], rtol=W_RTOL)
    assert np.all(actual[~normal] <= 2 * desired[~normal] + 1e-44) and np.all(desired[~normal] <= 2 * actual[~normal] + 1e-30)


def road_graph(n, n_chords, seed, integer):
    rng = random.Random(seed)
    edges = [(i, i + 1) for i in range(n - 1)]
    have = set(edges)
    while len(edges) < n - 1 + n_chords:
        a, b = rng.randrange(n), rng.randrange(n)
        if a == b or (a, b) in have or (b, a) in have:
            continue
        edges.append((a, b))
        have.add((a, b))
    d = [float(rng.randint(1, 6)) if integer else rng.uniform(3.0, 2900.0) for _ in edges]
    e = np.array(edges, dtype=np.int64)
    ue = np.concatenate([e, e[:, ::-1]])
    ud = np.concatenate([d, d])
    return torch.from_numpy(ue), torch.from_numpy(ud)


@pytest.mark.parametrize("name", ["small", "pems", "ties", "road400"])
def test_gpu_tables_match_reference(name):
    from mgadmm import gpu_graph as gg
    g = load_golden(f"g1_tables_{name}.npz")
    n, k, sigma = int(g["n"]), int(g["k"]), float(g["sigma"])
    ue, ud = torch.from_numpy(g["u_edges"]), torch.from_numpy(g["u_dist"])
    cl, dl = gg.k_nearest_neighbors(n, ue, ud, k)
    assert cl.dtype == torch.int32 and dl.dtype == torch.float32      # reference dtypes (utils.py:195-196)
    np.testing.assert_array_equal(cl.numpy(), g["knn_cl"])             # same neighbours in the same (tie) order
    np.testing.assert_array_equal(dl.numpy(), g["knn_dl"])             # float64 path sums rounded to float32
    cl = cl.to(torch.int64)
    from mgadmm.utils import _sigma_default
    sdef = _sigma_default(cl, dl)
    assert_weights_close(gg.undirected_graph_from_distance(cl, dl, sigma), g["knn_u_ew"], dl[:, 1:], sigma)
    assert_weights_close(gg.directed_graph_from_distance(cl, dl, sigma), g["knn_d_ew"], dl, sigma)
    assert_weights_close(gg.undirected_graph_from_distance(cl, dl), g["knn_u_ew_defsigma"], dl[:, 1:], sdef)
    assert_weights_close(gg.directed_graph_from_distance(cl, dl), g["knn_d_ew_defsigma"], dl, sdef)
    # physical adjacency tables: ragged rows, -1 / inf pads (quirk Q5: the pad index wraps, its weight is 0)
    pcl, pdl = torch.from_numpy(g["phys_cl"]), torch.from_numpy(g["phys_dl"])
    pu = gg.undirected_graph_from_distance(pcl, pdl, sigma).numpy()
    assert_weights_close(pu, g["phys_u_ew"], pdl[:, 1:], sigma)
    assert_weights_close(gg.directed_graph_from_distance(pcl, pdl, sigma), g["phys_d_ew"], pdl, sigma)
    assert np.all(pu[g["phys_cl"][:, 1:] == -1] == 0.0)


@pytest.mark.parametrize("integer", [False, True])
def test_gpu_knn_equals_host_search_on_a_larger_graph(integer):
    """5000-node road-like graph (the host builder is pinned to the reference by G1, ties included)."""
    from mgadmm import gpu_graph as gg, utils as mu
    n, k = 5000, 4
    ue, ud = road_graph(n, 900, seed=7, integer=integer)
    cl_h, dl_h = mu.k_nearest_neighbors(n, ue, ud, k)
    cl_g, dl_g = gg.k_nearest_neighbors(n, ue, ud, k)
    np.testing.assert_array_equal(cl_g.numpy(), cl_h.numpy())
    np.testing.assert_array_equal(dl_g.numpy(), dl_h.numpy())
    u_h = mu.undirected_graph_from_distance(cl_h.long(), dl_h)
    d_h = mu.directed_graph_from_distance(cl_h.long(), dl_h)
    u_g, d_g, (su, sd) = gg.weight_tables(cl_g.long(), dl_g)
    assert su == sd == mu._sigma_default(cl_h.long(), dl_h)
    assert_weights_close(u_g, u_h, dl_h[:, 1:], su)
    assert_weights_close(d_g, d_h, dl_h, sd)
    np.testing.assert_allclose(d_g.numpy().sum(1), 1.0, rtol=1e-6)     # rows of W_d are normalised


def test_gpu_knn_disconnected_duplicate_and_directed_edges():
    from mgadmm import gpu_graph as gg, utils as mu
    # two components (0-1-2, 3-4), a one-way edge 2->0, and a repeated edge whose later length wins
    edges = torch.tensor([[0, 1], [1, 0], [1, 2], [2, 1], [3, 4], [4, 3], [2, 0], [0, 1]])
    dists = torch.tensor([5.0, 5.0, 1.0, 1.0, 2.0, 2.0, 0.5, 3.0])
    cl, dl = gg.k_nearest_neighbors(5, edges, dists, 3)
    cl_h, dl_h = mu.k_nearest_neighbors(5, edges, dists, 3)
    np.testing.assert_array_equal(cl.numpy(), cl_h.numpy())
    np.testing.assert_array_equal(dl.numpy(), dl_h.numpy())
    assert cl[0].tolist() == [0, 1, 2, -1] and dl[0].tolist() == [0.0, 3.0, 4.0, float("inf")]
    assert cl[2].tolist() == [2, 0, 1, -1] and cl[3].tolist() == [3, 4, -1, -1]
    u, d, _ = gg.weight_tables(cl.long(), dl, u_sigma=2.0, d_sigma=2.0)
    assert float(u[3, 1]) == 0.0 and float(d[3, 2]) == 0.0 and abs(float(d[3].sum()) - 1.0) < 1e-6


def test_gpu_graph_errors():
    from mgadmm import _lib, gpu_graph as gg
    e = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(_lib.MgadmmError, match="negative edge length"):
        gg.k_nearest_neighbors(2, e, torch.tensor([1.0, -1.0]), 1)
    with pytest.raises(_lib.MgadmmError, match="out of range"):
        gg.k_nearest_neighbors(2, torch.tensor([[0, 2]]), torch.tensor([1.0]), 1)
    with pytest.raises(_lib.MgadmmError, match="exceeds"):
        gg.k_nearest_neighbors(2, e, torch.tensor([1.0, 1.0]), 40)
    with pytest.raises(ValueError):
        gg.k_nearest_neighbors(2, e, torch.tensor([1.0]), 1)
    # complete graph on 200 nodes: a search holds more live candidates than the kernel keeps -> loud error
    n = 200
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = ii != jj
    ee = torch.from_numpy(np.stack([ii[m], jj[m]], 1))
    dd = torch.from_numpy(1.0 + np.abs(ii[m] - jj[m]) * 1e-3)
    with pytest.raises(_lib.MgadmmError, match="live candidates"):
        gg.k_nearest_neighbors(n, ee, dd, 4)
    with pytest.raises(_lib.MgadmmError, match="default sigma"):
        gg.weight_tables(torch.tensor([[0, -1]]), torch.tensor([[0.0, float("inf")]]))


def test_solver_built_from_gpu_tables_matches_host_tables():
    """ADMM_algorithm(graph_backend='gpu'): same tables (to the weight tolerance) and the same solve."""
    import mgadmm
    g = load_golden("g1_tables_road400.npz")
    n = int(g["n"])
    gi = {"n_nodes": n, "u_edges": torch.from_numpy(g["u_edges"]), "u_dist": torch.from_numpy(g["u_dist"])}
    r = (n / 24) ** 0.5
    info = {"rho": 2 * r, "rho_u": 3 * r, "rho_d": 2 * r, "mu_u": 1, "mu_d1": 2, "mu_d2": 1}
    blks = {b: mgadmm.ADMM_algorithm(gi, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, graph_backend=b) for b in ("host", "gpu")}
    assert blks["gpu"].graph_backend == "gpu"
    assert torch.equal(blks["gpu"].connect_list, blks["host"].connect_list)
    assert_weights_close(blks["gpu"].d_ew[0], blks["host"].d_ew[0], blks["host"].dist_list, 50.0)
    y = 300 * torch.rand(3, 12, n, 1, generator=torch.Generator().manual_seed(1))
    xs = {}
    for b, blk in blks.items():
        blk.max_ADMM_iter = 5
        xs[b] = blk.combined_loop(y, print_info=False)
        blk.close()
    assert float((xs["gpu"] - xs["host"]).norm() / xs["host"].norm()) < 1e-5
