"""Graph construction on the GPU: the reference's table builders (utils.py:183-258) as HIP kernels behind
the C ABI (`mgadmm_knn_graph`, `mgadmm_weight_tables`, csrc/build.hip).  Same signatures and return types
as the functions of the same name in `mgadmm.utils` / the reference's `utils.py`; there is no CPU path
in here -- without the HIP library and a GPU these raise."""
import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["k_nearest_neighbors", "undirected_graph_from_distance", "directed_graph_from_distance", "weight_tables"]


def _dev_index(device):
    if device is None:
        return torch.cuda.current_device()
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def k_nearest_neighbors(n_nodes, edges, dists, k, device=None):
    """k+1 nearest nodes (self first) of every node by shortest-path distance (utils.py:183-204): one
    truncated Dijkstra per node, one GPU thread per search, networkx's visiting order (ties included).
    Returns (int32 (N,k+1) with -1 pads, float32 (N,k+1) with inf pads)."""
    e = np.ascontiguousarray(torch.as_tensor(edges).cpu().numpy(), dtype=np.int64).reshape(-1, 2)
    d = np.ascontiguousarray(torch.as_tensor(dists).cpu().numpy(), dtype=np.float64).reshape(-1)
    if len(e) != len(d):
        raise ValueError(f"{len(e)} edges but {len(d)} distances")
    nn = np.empty((n_nodes, k + 1), dtype=np.int32)
    nd = np.empty((n_nodes, k + 1), dtype=np.float32)
    _lib.check(_lib.lib.mgadmm_knn_graph(int(n_nodes), len(e), e.ctypes.data_as(C.POINTER(C.c_int64)),
                                         d.ctypes.data_as(C.POINTER(C.c_double)), int(k),
                                         nn.ctypes.data_as(C.POINTER(C.c_int32)), nd.ctypes.data_as(C.POINTER(C.c_float)),
                                         _dev_index(device)))
    return torch.from_numpy(nn), torch.from_numpy(nd)


def weight_tables(connect_list, dist_list, u_sigma=None, d_sigma=None, regularized=True, device=None, which="ud"):
    """Both weight tables in one call.  Returns (u_ew (N,k) | None, d_ew (N,k+1) | None, (u_sigma, d_sigma))."""
    cl = np.ascontiguousarray(torch.as_tensor(connect_list).cpu().numpy(), dtype=np.int64)
    dl = np.ascontiguousarray(torch.as_tensor(dist_list).cpu().numpy(), dtype=np.float32)
    if cl.shape != dl.shape or cl.ndim != 2:
        raise ValueError(f"connect_list {cl.shape} and dist_list {dl.shape} must be equal (N, k+1) tables")
    n, k1 = cl.shape
    u = np.empty((n, k1 - 1), dtype=np.float32) if "u" in which else None
    dd = np.empty((n, k1), dtype=np.float32) if "d" in which else None
    sig = np.zeros(2, dtype=np.float64)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.lib.mgadmm_weight_tables(n, k1, cl.ctypes.data_as(C.POINTER(C.c_int64)), dl.ctypes.data_as(fp),
                                             float(u_sigma) if u_sigma is not None else 0.0,
                                             float(d_sigma) if d_sigma is not None else 0.0, int(bool(regularized)),
                                             u.ctypes.data_as(fp) if u is not None else None,
                                             dd.ctypes.data_as(fp) if dd is not None else None,
                                             sig.ctypes.data_as(C.POINTER(C.c_double)), _dev_index(device)))
    return (torch.from_numpy(u) if u is not None else None, torch.from_numpy(dd) if dd is not None else None,
            (float(sig[0]), float(sig[1])))


def undirected_graph_from_distance(connect_list, dist_list, u_sigma=None, regularized=True, device=None):
    """Undirected weights (N,k) float32 (utils.py:206-238)."""
    return weight_tables(connect_list, dist_list, u_sigma=u_sigma, regularized=regularized, device=device, which="u")[0]


def directed_graph_from_distance(connect_list, dist_list, d_sigma=None, regularized=True, device=None):
    """Directed (temporal) weights (N,k+1) float32, rows normalised (utils.py:240-258)."""
    return weight_tables(connect_list, dist_list, d_sigma=d_sigma, regularized=regularized, device=device, which="d")[1]
