"""Drop-in for the reference's stand-alone ``CG_script.conjugate_gradient`` (CG_script.py:3-44).

The dense system A x = b is handed to the same HIP conjugate-gradient kernels that serve
``ADMM_algorithm.CG_solver``: A is expressed as the "Laplacian" I - W with W = I - A stored as CSR,
on a two-slice time axis whose second slice is identically zero (it contributes exactly 0 to every
dot product, so alpha, beta and the iterates are those of CG on A alone), in float64.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ADMM import _device, _ptr, _stream_ptr
from .graph import Graph


def conjugate_gradient(A, b, x0=None, tol=1e-10, max_iter=1000):
    """Solve A x = b by CG.  Returns (x, n_iter) like the reference: n_iter = k+1 at convergence
    (sqrt(r.r) < tol), max_iter otherwise."""
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    n = b.shape[0]
    assert A.shape == (n, n), "A must be (n, n) with n = len(b)"
    W = np.eye(n) - A
    rowptr = np.arange(0, n * n + 1, n, dtype=np.int32)
    col = np.tile(np.arange(n, dtype=np.int32), n)
    # float32 CSR values would round A: carry the value in two float32 terms (hi + lo), duplicates sum
    hi = W.astype(np.float32)
    lo = (W - hi.astype(np.float64)).astype(np.float32)
    rowptr2 = (rowptr * 2).astype(np.int32)
    col2 = np.concatenate([col.reshape(n, n), col.reshape(n, n)], 1).reshape(-1).astype(np.int32)
    val2 = np.concatenate([hi, lo], 1).reshape(-1).astype(np.float32)
    dev = _device()
    ident = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.zeros(n, dtype=np.float32))
    g = Graph(n, 2, (rowptr2, col2, val2), ident, device=dev.index or 0)
    p = _lib.Params()
    p.rho = p.rho_d = p.mu_d1 = p.mu_d2 = 1.0
    p.rho_u, p.mu_u = 0.0, 1.0            # LHS_zu = mu_u * (I - W) + rho_u/2 * I = A
    p.t_in, p.ablation = 1, 0
    p.cg_tol, p.max_cg_iter = float(tol), int(max_iter)
    p.admm_tol, p.max_admm_iter = 1e-6, 1
    p.dtype, p.check_stop, p.path, p.record_cg_coeffs = _lib.F64, 0, _lib.PATH_STREAM, 0
    h = C.c_void_p()
    _lib.check(_lib.lib.mgadmm_solver_create(g.handle, C.byref(p), 1, C.byref(h)))
    try:
        rhs = torch.zeros(1, 2, n, dtype=torch.float64, device=dev)
        rhs[0, 0] = torch.from_numpy(b).to(dev)
        x0d = None
        if x0 is not None:
            x0d = torch.zeros_like(rhs)
            x0d[0, 0] = torch.from_numpy(np.asarray(x0, dtype=np.float64).reshape(-1)).to(dev)
        out = torch.empty_like(rhs)
        iters = np.zeros(1, dtype=np.int32)
        _lib.check(_lib.lib.mgadmm_cg(h, _lib.LHS_ZU, _ptr(rhs), _ptr(x0d), None, _ptr(out),
                                      iters.ctypes.data_as(C.POINTER(C.c_int32)), None, None, 1, _stream_ptr(dev)))
        x = out[0, 0].cpu().numpy()
    finally:
        _lib.lib.mgadmm_solver_destroy(h)
        g.close()
    n_iter = int(iters[0])
    return x, (n_iter if n_iter > 0 else max_iter)
