"""Device graph handles: reference-style padded tables -> CSR -> libmgadmm graph.

CSR mapping (reference ADMM.py:143-148, 166-177, 197-223):
  W_u : row i, columns connect_list[i, 1..k] (!= -1), values u_ew[i, :]
  W_d : row i, columns connect_list[i, 0..k] (!= -1), values d_ew[i, :]
The exact transpose of W_d is built inside the library (no atomics, no scatter_add).
C > 1 channels are folded into the node axis: node (n, c) -> n*C + c, every edge replicated per
channel, so a (B,T,N,C) tensor is a (B,T,N*C) tensor on the expanded graph with no data movement.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def tables_to_csr(connect_list, weights, first_col):
    """Rows of a padded table -> (rowptr int32, col int32, val float32); pads (-1) dropped."""
    cl = np.asarray(torch.as_tensor(connect_list).cpu().numpy())[:, first_col:]
    w = np.asarray(torch.as_tensor(weights).cpu().numpy(), dtype=np.float32)
    assert cl.shape == w.shape, f"table shapes differ: {cl.shape} vs {w.shape}"
    keep = cl != -1
    if (cl[keep] < 0).any() or (cl[keep] >= cl.shape[0]).any():
        raise ValueError("Index out of bounds")           # reference ADMM.py:205-206
    rowptr = np.zeros(cl.shape[0] + 1, dtype=np.int32)
    np.cumsum(keep.sum(1), out=rowptr[1:])
    return rowptr, cl[keep].astype(np.int32), w[keep].astype(np.float32)


def expand_channels(csr, C_):
    """Block-replicate a CSR matrix per channel: (i,j,w) -> (i*C+c, j*C+c, w)."""
    if C_ == 1:
        return csr
    rowptr, col, val = csr
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    new_deg = np.repeat(deg, C_)
    new_rowptr = np.zeros(n * C_ + 1, dtype=np.int32)
    np.cumsum(new_deg, out=new_rowptr[1:])
    new_col = np.empty(len(col) * C_, dtype=np.int32)
    new_val = np.empty(len(col) * C_, dtype=np.float32)
    for i in range(n):
        seg = slice(rowptr[i], rowptr[i + 1])
        for c in range(C_):
            o = new_rowptr[i * C_ + c]
            new_col[o:o + deg[i]] = col[seg] * C_ + c
            new_val[o:o + deg[i]] = val[seg]
    return new_rowptr, new_col, new_val


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Graph:
    """Owns one mgadmm_graph handle (device-resident CSR of W_u, W_d, W_d^T)."""

    def __init__(self, n_nodes, T, u_csr, d_csr=None, *, band_w=None, skip=1, transpose_by_gather=False,
                 q1_identity_t0=True, reorder=False, device=0):
        self.n_nodes, self.T, self.device = int(n_nodes), int(T), int(device)
        self._keep = []
        d = _lib.GraphDesc()
        d.n_nodes, d.T = self.n_nodes, self.T
        ur, uc, uv = [np.ascontiguousarray(a) for a in u_csr]
        self._keep += [ur, uc, uv]
        d.u_rowptr, d.u_col, d.u_val = _ptr(ur, C.c_int32), _ptr(uc, C.c_int32), _ptr(uv, C.c_float)
        if d_csr is not None:
            dr, dc, dv = [np.ascontiguousarray(a) for a in d_csr]
            self._keep += [dr, dc, dv]
            d.temporal_mode = _lib.TEMPORAL_SPATIAL
            d.d_rowptr, d.d_col, d.d_val = _ptr(dr, C.c_int32), _ptr(dc, C.c_int32), _ptr(dv, C.c_float)
        else:
            bw = np.ascontiguousarray(np.asarray(band_w, dtype=np.float32).reshape(self.T, int(skip)))
            self._keep.append(bw)
            d.temporal_mode = _lib.TEMPORAL_BAND
            d.skip = int(skip)
            d.band_w = _ptr(bw, C.c_float)
        d.transpose_by_gather = int(bool(transpose_by_gather))
        d.q1_identity_t0 = int(bool(q1_identity_t0))
        d.reorder = int(reorder)             # 0 none, 1 RCM, 2 greedy cluster order
        d.device = self.device
        h = C.c_void_p()
        _lib.check(_lib.lib.mgadmm_graph_create(C.byref(d), C.byref(h)))
        self.handle = h
        self.spatial = d_csr is not None

    def transpose_csr(self):
        nnz = C.c_int32()
        _lib.check(_lib.lib.mgadmm_graph_transpose_nnz(self.handle, C.byref(nnz)))
        rp = np.zeros(self.n_nodes + 1, dtype=np.int32)
        col = np.zeros(max(1, nnz.value), dtype=np.int32)
        val = np.zeros(max(1, nnz.value), dtype=np.float32)
        _lib.check(_lib.lib.mgadmm_graph_get_transpose(self.handle, _ptr(rp, C.c_int32), _ptr(col, C.c_int32),
                                                       _ptr(val, C.c_float)))
        return rp, col[:nnz.value], val[:nnz.value]

    def perm(self):
        p = np.zeros(self.n_nodes, dtype=np.int32)
        _lib.check(_lib.lib.mgadmm_graph_get_perm(self.handle, _ptr(p, C.c_int32)))
        return p

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib.mgadmm_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
