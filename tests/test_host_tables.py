"""Host-side graph construction of the product package (mgadmm.utils / mgadmm.graph) against the
golden tables produced by the reference (G1) and its literal KATs.  No GPU needed."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from mgadmm import utils as mu
from mgadmm.dist import shard_bounds
from mgadmm.graph import expand_channels, tables_to_csr


def test_connect_list_kat():
    # reference utils.py:297-300
    edges = torch.tensor([[0, 1], [1, 2], [2, 3], [3, 2], [2, 1], [1, 0]], dtype=torch.int)
    cl, dl = mu.connect_list(4, edges, torch.tensor([1, 2, 3, 3, 2, 1]))
    assert cl.dtype == torch.int64
    assert cl.tolist() == [[0, 1, -1], [1, 0, 2], [2, 1, 3], [3, 2, -1]]
    inf = float("inf")
    assert dl.tolist() == [[0, 1, inf], [0, 1, 2], [0, 2, 3], [0, 3, inf]]


@pytest.mark.parametrize("name", ["small", "pems", "ties", "road400"])
def test_tables_match_reference(name):
    g = load_golden(f"g1_tables_{name}.npz")
    n, k, sigma = int(g["n"]), int(g["k"]), float(g["sigma"])
    ue, ud = torch.from_numpy(g["u_edges"]), torch.from_numpy(g["u_dist"])
    cl, dl = mu.k_nearest_neighbors(n, ue, ud, k)
    assert cl.dtype == torch.int32 and dl.dtype == torch.float32      # reference dtypes (utils.py:195-196)
    assert np.array_equal(cl.numpy(), g["knn_cl"])
    np.testing.assert_array_equal(dl.numpy(), g["knn_dl"])
    cl = cl.to(torch.int64)
    # same torch float32 kernels as the reference -> bit-identical weights
    np.testing.assert_array_equal(mu.undirected_graph_from_distance(cl, dl, sigma).numpy(), g["knn_u_ew"])
    np.testing.assert_array_equal(mu.directed_graph_from_distance(cl, dl, sigma).numpy(), g["knn_d_ew"])
    np.testing.assert_array_equal(mu.undirected_graph_from_distance(cl, dl).numpy(), g["knn_u_ew_defsigma"])
    np.testing.assert_array_equal(mu.directed_graph_from_distance(cl, dl).numpy(), g["knn_d_ew_defsigma"])
    pcl, pdl = mu.connect_list(n, ue, ud)
    assert np.array_equal(pcl.numpy(), g["phys_cl"]) and np.array_equal(pdl.numpy(), g["phys_dl"])
    np.testing.assert_array_equal(mu.undirected_graph_from_distance(pcl, pdl, sigma).numpy(), g["phys_u_ew"])
    np.testing.assert_array_equal(mu.directed_graph_from_distance(pcl, pdl, sigma).numpy(), g["phys_d_ew"])


def test_skip_tables_match_notebook_kat():
    # directed_graph.ipynb: T=5, skip=2
    d_ew, tl = mu.skip_connection_tables(3, 5, 2)
    assert d_ew.shape == (5, 2, 3)
    assert d_ew[:, :, 0].tolist() == [[0, 0], [1, 0], [.5, .5], [.5, .5], [.5, .5]]
    assert tl.tolist() == [[-1, -2], [0, -1], [1, 0], [2, 1], [3, 2]]
    g = load_golden("g6_kats.npz")
    np.testing.assert_array_equal(d_ew.numpy(), g["line2_d_ew"])
    np.testing.assert_array_equal(tl.numpy(), g["line2_time_list"])


def test_tables_to_csr_drops_pads_and_keeps_order():
    cl = torch.tensor([[0, 2, -1], [1, 0, 2], [2, -1, -1]])
    w = torch.tensor([[.5, 0.], [.25, .75], [0., 0.]])
    rp, col, val = tables_to_csr(cl, w, 1)
    assert rp.tolist() == [0, 1, 3, 3] and col.tolist() == [2, 0, 2] and val.tolist() == [.5, .25, .75]
    rp, col, val = tables_to_csr(cl, torch.ones(3, 3), 0)
    assert rp.tolist() == [0, 2, 5, 6] and col.tolist() == [0, 2, 1, 0, 2, 2]
    with pytest.raises(ValueError, match="Index out of bounds"):      # reference ADMM.py:205-206
        tables_to_csr(torch.tensor([[0, 7]]), torch.ones(1, 1), 1)


def test_expand_channels_is_block_replication():
    rp, col, val = tables_to_csr(torch.tensor([[0, 1], [1, 0]]), torch.tensor([[.5], [.25]]), 1)
    rp2, col2, val2 = expand_channels((rp, col, val), 3)
    assert rp2.tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert col2.tolist() == [3, 4, 5, 0, 1, 2]
    assert val2.tolist() == [.5, .5, .5, .25, .25, .25]


def test_knn_from_points_self_first():
    rng = np.random.default_rng(0)
    cl, dl = mu.knn_from_points(rng.random((200, 2)), 4, scale=1000.0)
    assert cl.shape == (200, 5) and (cl[:, 0] == torch.arange(200)).all() and (dl[:, 0] == 0).all()
    assert (dl[:, 1:] > 0).all()


def test_shard_bounds_cover_the_batch():
    for B in (1, 7, 8, 4096, 2049):
        for ws in (1, 2, 3, 8):
            cuts = [shard_bounds(B, ws, r) for r in range(ws)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
