"""Graph-table construction with the reference's function names and return conventions
(reference: utils.py:39-52, 144-258, 294-295).  Host-side set-up code: it runs once per graph and
feeds `mgadmm.graph.build_csr`; the per-iteration work happens in the HIP kernels.

Differences from the reference, all deliberate:
  * `k_nearest_neighbors` is a Dijkstra truncated after k+1 settled nodes (O(N k log N) instead of
    the reference's all-pairs O(N^2)); it settles nodes in networkx's order, so the tables are
    identical (pinned by tests/golden/g1_tables_*.npz).
  * nothing prints.
"""
import heapq

import numpy as np
import torch

__all__ = ["physical_graph", "get_data_difference", "connect_list", "k_nearest_neighbors",
           "undirected_graph_from_distance", "directed_graph_from_distance", "expand_time_dimension",
           "skip_connection_tables", "knn_from_points"]


def physical_graph(df, sensor_dict=None):
    """Bidirectional edge list from a distance table with 'from'/'to' columns and the distance in
    the last column (utils.py:39-52).  Returns (n_edges, u_edges (2E,2) int64, u_distance (2E,))."""
    src = list(df["from"].values)
    dst = list(df["to"].values)
    if sensor_dict is not None:
        src = [sensor_dict[i] for i in src]
        dst = [sensor_dict[i] for i in dst]
    n_edges = len(src)
    u_edges = torch.tensor([src + dst, dst + src]).T
    w = torch.tensor(df[df.columns[-1]].values)
    return n_edges, u_edges, torch.cat([w, w])


def get_data_difference(data):
    """First difference along time of a (B,T,N,C) tensor (utils.py:144-153)."""
    assert data.ndim == 4, "Data should have 4 dims (B, T, N, C)"
    return data[:, 1:] - data[:, :-1]


def connect_list(n_nodes, edges, dists):
    """Padded physical adjacency (utils.py:156-181): column 0 = the node itself, neighbours filled
    from the last slot downwards in edge order, pads -1 / inf.  Returns (int64 (N,kmax+1), float32)."""
    e = np.asarray(torch.as_tensor(edges).cpu().numpy(), dtype=np.int64)
    d = torch.as_tensor(dists).cpu().numpy()
    counts = np.bincount(e[:, 0], minlength=n_nodes).astype(np.int64)
    k = int(counts.max())
    cl = -np.ones((n_nodes, k + 1), dtype=np.int64)
    dl = np.full((n_nodes, k + 1), np.inf, dtype=np.float32)
    slot = counts.copy()
    for i in range(len(e)):
        s = e[i, 0]
        cl[s, slot[s]] = e[i, 1]
        dl[s, slot[s]] = d[i]
        slot[s] -= 1
    cl[:, 0] = np.arange(n_nodes)
    dl[:, 0] = 0
    return torch.from_numpy(cl), torch.from_numpy(dl)


def k_nearest_neighbors(n_nodes, edges, dists, k):
    """k+1 nearest nodes (self first) by shortest-path distance (utils.py:183-204).
    Returns (int32 (N,k+1) with -1 pads, float32 (N,k+1) with inf pads) like the reference."""
    e = np.asarray(torch.as_tensor(edges).cpu().numpy(), dtype=np.int64)
    d = np.asarray(torch.as_tensor(dists).cpu().numpy(), dtype=np.float64)
    adj = [dict() for _ in range(n_nodes)]
    for i in range(len(e)):
        adj[int(e[i, 0])][int(e[i, 1])] = float(d[i])
    nn = -np.ones((n_nodes, k + 1), dtype=np.int32)
    nd = np.full((n_nodes, k + 1), np.inf, dtype=np.float32)
    for src in range(n_nodes):
        done = {}
        best = {src: 0.0}
        tick = 0
        heap = [(0.0, 0, src)]
        while heap and len(done) < k + 1:
            dist, _, v = heapq.heappop(heap)
            if v in done:
                continue
            done[v] = dist
            for u, w in adj[v].items():
                if u in done:
                    continue
                cand = dist + w
                if u not in best or cand < best[u]:
                    best[u] = cand
                    tick += 1
                    heapq.heappush(heap, (cand, tick, u))
        for j, (node, dist) in enumerate(sorted(done.items(), key=lambda kv: kv[1])):
            nn[src, j] = node
            nd[src, j] = dist
    return torch.from_numpy(nn), torch.from_numpy(nd)


def knn_from_points(points, k, scale=1.0):
    """Euclidean k-NN tables for synthetic graphs (BASELINE configs 3/4): (int64 (N,k+1), float32).
    Not in the reference (its kNN is graph-distance Dijkstra); same table format."""
    from scipy.spatial import cKDTree
    pts = np.asarray(points, dtype=np.float64)
    dist, idx = cKDTree(pts).query(pts, k=k + 1)
    idx = idx.astype(np.int64)
    # make sure column 0 is the node itself even with duplicate points
    rows = np.arange(len(pts))
    bad = idx[:, 0] != rows
    for r in np.nonzero(bad)[0]:
        j = np.nonzero(idx[r] == r)[0]
        if len(j):
            idx[r, [0, j[0]]] = idx[r, [j[0], 0]]
            dist[r, [0, j[0]]] = dist[r, [j[0], 0]]
        else:
            idx[r, 0] = r
            dist[r, 0] = 0.0
    return torch.from_numpy(idx), torch.from_numpy((dist * scale).astype(np.float32))


def _sigma_default(cl, dl):
    m = (cl != -1) & (dl != 0)
    v = dl[m]
    return max(v.max().item() / 50, v.min().item() * 50)


def undirected_graph_from_distance(connect_list, dist_list, u_sigma=None, regularized=True):
    """Undirected weights (N,k) float32 on columns 1..k: exp(-d/sigma), 0 at pads, divided by
    sqrt(deg_i * deg_j) (utils.py:206-238; a pad index wraps to the last node but carries weight 0)."""
    cl = torch.as_tensor(connect_list)
    dl = torch.as_tensor(dist_list)
    if u_sigma is None:
        u_sigma = _sigma_default(cl, dl)
    nb = cl[:, 1:]
    w = torch.exp(-dl[:, 1:] / u_sigma)
    w = torch.where(nb == -1, torch.zeros_like(w), w)
    if regularized:
        deg = w.sum(1)
        prod = deg[:, None] * deg[nb.long()]
        w = w * torch.where(prod > 0, 1 / torch.sqrt(prod), torch.zeros_like(prod))
    return w


def directed_graph_from_distance(connect_list, dist_list, d_sigma=None, regularized=True):
    """Directed (temporal) weights (N,k+1) float32 incl. the self column, row-normalised
    (utils.py:240-258)."""
    cl = torch.as_tensor(connect_list)
    dl = torch.as_tensor(dist_list)
    if d_sigma is None:
        d_sigma = _sigma_default(cl, dl)
    w = torch.exp(-dl / d_sigma)
    w = torch.where(cl == -1, torch.zeros_like(w), w)
    if regularized:
        deg = w.sum(1, keepdim=True)
        w = w * torch.where(deg > 0, 1 / deg, torch.zeros_like(deg))
    return w


def expand_time_dimension(ew, T):
    """(N,k) -> (T,N,k) by repetition (utils.py:294-295)."""
    return ew.unsqueeze(0).repeat(T, 1, 1)


def skip_connection_tables(n_nodes, T, skip):
    """Line-graph temporal tables of ADMM.py:41-52: d_ew (T,skip,N) float32 with row t holding
    1/min(t,skip) on s < min(t,skip), and time_list (T,skip) = t - (s+1)."""
    w = torch.zeros(T, skip)
    for t in range(1, T):
        m = min(t, skip)
        w[t, :m] = 1.0 / m
    d_ew = w[:, :, None].repeat(1, 1, n_nodes)
    time_list = torch.arange(0, T).unsqueeze(1) - torch.arange(1, skip + 1)
    return d_ew, time_list
