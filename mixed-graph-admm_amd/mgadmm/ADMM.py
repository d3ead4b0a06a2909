"""Drop-in host for the reference's ``ADMM.py``: same class name, constructor, attributes and
``combined_loop(y, mask) -> x`` contract (reference ADMM.py:11-648), with the per-iteration work --
three batched conjugate-gradient solves whose matvec is the sparse mixed-graph Laplacian, the
soft-threshold prox, the dual updates and the residual history -- executed by hand-written gfx950
HIP kernels behind the C ABI of ``include/mgadmm.h`` (loaded with ctypes, see ``_lib.py``).

There is NO CPU fallback: tensors are moved to the MI355X, and every method below raises if the
HIP library or the GPU is missing.

Batched semantics: B samples are B independent reference runs (per-sample CG convergence; the
reference itself cannot run B > 1, ADMM.py:362).  Whole-batch residual norms are the Frobenius norms
over the batch tensor like ADMM.py:612-636.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from . import utils as _u
from .graph import Graph, expand_channels, tables_to_csr

__all__ = ["ADMM_algorithm", "initial_guess", "initial_interpolation"]

_TORCH2MG = {torch.float32: _lib.F32, torch.float64: _lib.F64}
_NP = {torch.float32: np.float32, torch.float64: np.float64}


def _device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("mgadmm needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU path")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _stream_ptr(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class ADMM_algorithm():
    """MI355X-native mixed-graph ADMM solver with the reference's interface (ADMM.py:15).

    Extra keyword-only arguments (not in the reference):
      device          torch device of the GPU to use (default: current cuda device)
      compute_dtype   torch.float32 (default, the HIP fast path) or torch.float64, or 'match' to
                      compute in the dtype of ``y``
      bug_compat      reproduce the reference's quirk Q1 (Ldr_T keeps the identity on the t=0 block,
                      ADMM.py:221-222); default True so iterates match the reference
      tables          optional (connect_list, dist_list) to skip neighbour search (large graphs)
      reorder         internal node order: False/0, 'rcm'/1, 'cluster'/2 (greedy cluster growth; enables the
                      LDS-tiled SpMM kernel and the fused Ldr^T Ldr kernel) or 'auto' (cluster order for every graph
                      beyond the LDS-resident path, N > 512)
      record_cg_coeffs  keep alpha/beta of every CG iteration: True/False/'auto' (B <= 64)
      graph_backend   where the kNN search and the weight tables are computed: 'host' (NumPy, like the
                      reference's set-up code), 'gpu' (mgadmm.gpu_graph: HIP kernels) or 'auto' (gpu for
                      N >= 2048, where the host search takes seconds to minutes)
      cg_convergence  'per_sample' (default): every sample of a batch stops its CG on its own residual, i.e. B samples
                      are B independent reference runs; 'batch_max': the reference's literal test
                      ``sqrt(rr).max() < CG_tol`` (ADMM.py:360) -- all samples iterate until the largest residual of
                      the batch is below the tolerance (streaming kernels only; one count per solve)
    """

    def __init__(self, graph_info, ADMM_info, use_kNN=False, k=4, u_sigma=None, d_sigma=None, expand_time_dim=True,
                 ablation='None', t_in=12, T=24, use_line_graph=False, skip_connection=1, *, device=None,
                 compute_dtype=torch.float32, bug_compat=True, tables=None, reorder='auto', record_cg_coeffs='auto',
                 path='auto', graph_backend='auto', cg_convergence='per_sample'):
        if cg_convergence not in ('per_sample', 'batch_max'):
            raise ValueError(f"cg_convergence must be 'per_sample' or 'batch_max', got {cg_convergence!r}")
        self.cg_convergence = cg_convergence
        if graph_backend not in ('auto', 'host', 'gpu'):
            raise ValueError(f"graph_backend must be 'auto', 'host' or 'gpu', got {graph_backend!r}")
        if graph_backend == 'auto':
            graph_backend = 'gpu' if graph_info['n_nodes'] >= 2048 else 'host'
        self.graph_backend = graph_backend
        if graph_backend == 'gpu':
            from . import gpu_graph as _g
        else:
            _g = _u
        self._tables_mod = _g
        self.t_in = t_in
        self.T = T
        self.use_line_graph = use_line_graph
        self.skip_connection = skip_connection
        self.n_nodes = graph_info['n_nodes']
        self.u_edges = graph_info.get('u_edges')
        self.u_dists = graph_info.get('u_dist')
        self.use_kNN = use_kNN
        if tables is not None:
            self.connect_list, self.dist_list = torch.as_tensor(tables[0]).to(torch.int64), torch.as_tensor(tables[1]).float()
        elif use_kNN:
            self.connect_list, self.dist_list = _g.k_nearest_neighbors(self.n_nodes, self.u_edges, self.u_dists, k)
            self.connect_list = self.connect_list.to(torch.int64)
        else:
            self.connect_list, self.dist_list = _u.connect_list(self.n_nodes, self.u_edges, self.u_dists)

        assert ablation in ['None', 'DGTV', 'DGLR', 'UT'], "ablation should be in ['None', 'DGTV', 'DGLR', 'UT']"
        self.ablation = ablation
        self.u_ew = _g.undirected_graph_from_distance(self.connect_list, self.dist_list, u_sigma=u_sigma)
        if expand_time_dim:
            self.u_ew = _u.expand_time_dimension(self.u_ew, T)
        if not use_line_graph:
            self.d_ew = _g.directed_graph_from_distance(self.connect_list, self.dist_list, d_sigma=d_sigma)
            if expand_time_dim:
                self.d_ew = _u.expand_time_dimension(self.d_ew, T - 1)
        else:
            self.d_ew, self.time_list = _u.skip_connection_tables(self.n_nodes, T, skip_connection)

        self.rho, self.rho_u, self.rho_d = ADMM_info['rho'], ADMM_info['rho_u'], ADMM_info['rho_d']
        self.mu_u, self.mu_d1, self.mu_d2 = ADMM_info['mu_u'], ADMM_info['mu_d1'], ADMM_info['mu_d2']

        self.max_CG_iter = 100          # ADMM.py:76-80
        self.max_inner_iter = 100
        self.CG_tol = 1e-8
        self.ADMM_tol = 1e-6
        self.max_ADMM_iter = 150

        self._device_arg = device
        self.compute_dtype = compute_dtype
        self.bug_compat = bug_compat
        self.reorder = reorder
        self.record_cg_coeffs = record_cg_coeffs
        self.path = path
        self.check_stop = True
        self._graphs = {}      # C -> (Graph, table identity)
        self._solvers = {}     # (C, dtype) -> [handle, Bmax]
        self._reset_history()
        self._set_res_name()

    # ------------------------------------------------------------------ bookkeeping
    def _set_res_name(self):
        self.res_name = ['zu']
        if self.ablation in ['None', 'DGLR']:
            self.res_name.append('phi')
        if self.ablation != 'DGLR':
            self.res_name.append('zd')

    def _reset_history(self):
        self.alpha_x, self.beta_x, self.alpha_zu, self.beta_zu, self.alpha_zd, self.beta_zd = [], [], [], [], [], []
        self.CG_iter_x, self.CG_iter_zu, self.CG_iter_zd = [], [], []
        self.p_res_list, self.d_res_list, self.x_shift_list, self.delta_x_per_step = [], [], [], []
        self.DGTV_list, self.DGLR_list, self.GLR_list, self.recover_list = [], [], [], []

    def init_iterations(self, ablation, use_line_graph=False):
        """Reset the history and switch ablation / temporal graph (ADMM.py:100-133).  Like the reference: the history
        lists are cleared EXCEPT ``recover_list`` (the reference never resets it, ADMM.py:100-133), the line graph is
        rebuilt with skip_connection = 1 (ADMM.py:104-106: ``ones((N, T, 1))`` whatever the constructor's skip was)
        and the kNN-directed weights are rebuilt with the default sigma (ADMM.py:108: no ``d_sigma`` argument)."""
        keep_recover = self.recover_list
        if use_line_graph:
            self.use_line_graph = True
            self.skip_connection = 1
            self.d_ew, self.time_list = _u.skip_connection_tables(self.n_nodes, self.T, 1)
        else:
            self.use_line_graph = False
            self.d_ew = _u.expand_time_dimension(
                self._tables_mod.directed_graph_from_distance(self.connect_list, self.dist_list, d_sigma=None), self.T - 1)
        assert ablation in ['None', 'DGTV', 'DGLR', 'UT']
        self.ablation = ablation
        self._reset_history()
        self.recover_list = keep_recover
        self._set_res_name()
        self.close()

    @property
    def device(self):
        return _device(self._device_arg)

    def close(self):
        for h, _ in self._solvers.values():
            _lib.lib.mgadmm_solver_destroy(h)
        self._solvers = {}
        for gph, _ in self._graphs.values():
            gph.close()
        self._graphs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ handles
    def _table_key(self):
        # identity AND in-place version of every table: `blk.u_ew = ...` and `blk.u_ew.mul_(2)` both rebuild the device CSR
        tv = lambda t: (id(t), getattr(t, "_version", 0), tuple(t.shape))
        return (tv(self.connect_list), tv(self.u_ew), tv(self.d_ew), self.use_line_graph, self.skip_connection,
                self.use_kNN, self.bug_compat, self.T)

    @staticmethod
    def _time_invariant(tab, name):
        """The device CSR holds ONE spatial slice: the reference's (T, N, k) tables are pure repeats of it
        (utils.py:294-295).  A table that really varies over time would be silently collapsed -- refuse it."""
        if tab.dim() == 3:
            if tab.shape[0] > 1 and not bool((tab == tab[0:1]).all()):
                raise NotImplementedError(f"{name} varies over the time axis; only time-expanded repeats of one (N, k) table "
                                          "(expand_time_dimension, utils.py:294-295) are supported")
            return tab[0]
        return tab

    def _graph(self, Cn):
        key = self._table_key()
        ent = self._graphs.get(Cn)
        if ent is not None and ent[1] == key:
            return ent[0]
        if ent is not None:                     # tables were replaced after construction: rebuild
            for k2 in [k2 for k2 in self._solvers if k2[0] == Cn]:
                _lib.lib.mgadmm_solver_destroy(self._solvers.pop(k2)[0])
            ent[0].close()
        u_ew = self._time_invariant(self.u_ew, "u_ew")
        u_csr = expand_channels(tables_to_csr(self.connect_list, u_ew, 1), Cn)
        N = self.n_nodes * Cn
        reorder = self.reorder
        if reorder == 'auto':
            # greedy cluster order (csrc/graph.hip) for every graph the LDS-resident path does not take: the tiled / fused
            # streaming kernels need it (N = 600 ... 1000 at B = 4096: +18 % over the plain row kernels, tools/size_sweep.py)
            reorder = 2 if N > 512 else 0
        reorder = {'rcm': 1, 'cluster': 2}[reorder] if isinstance(reorder, str) else int(reorder)
        dev = self.device
        if self.use_line_graph:
            bw = self.d_ew[:, :, 0].contiguous().cpu().numpy()
            gph = Graph(N, self.T, u_csr, None, band_w=bw, skip=self.skip_connection, reorder=reorder,
                        device=dev.index or 0)
        else:
            d_ew = self._time_invariant(self.d_ew, "d_ew")
            d_csr = expand_channels(tables_to_csr(self.connect_list, d_ew, 0), Cn)
            gph = Graph(N, self.T, u_csr, d_csr, transpose_by_gather=not self.use_kNN,
                        q1_identity_t0=self.bug_compat, reorder=reorder, device=dev.index or 0)
        self._graphs[Cn] = (gph, key)
        return gph

    def _params(self, dtype, B):
        p = _lib.Params()
        p.rho, p.rho_u, p.rho_d = float(self.rho), float(self.rho_u), float(self.rho_d)
        p.mu_u, p.mu_d1, p.mu_d2 = float(self.mu_u), float(self.mu_d1), float(self.mu_d2)
        p.t_in = int(self.t_in)
        p.ablation = _lib.ABLATIONS[self.ablation]
        p.cg_tol, p.max_cg_iter = float(self.CG_tol), int(self.max_CG_iter)
        p.admm_tol, p.max_admm_iter = float(self.ADMM_tol), int(self.max_ADMM_iter)
        p.dtype = _TORCH2MG[dtype]
        p.check_stop = int(bool(self.check_stop))
        p.path = {'auto': _lib.PATH_AUTO, 'stream': _lib.PATH_STREAM, 'lds': _lib.PATH_LDS}[self.path]
        rec = self.record_cg_coeffs
        if rec == 'auto':
            rec = B <= 64
        p.record_cg_coeffs = int(bool(rec))
        p.cg_convergence = _lib.CG_BATCH_MAX if self.cg_convergence == 'batch_max' else _lib.CG_PER_SAMPLE
        p.max_inner_iter = int(self.max_inner_iter)
        return p

    def _solver(self, Cn, dtype, B):
        gph = self._graph(Cn)
        p = self._params(dtype, B)
        ent = self._solvers.get((Cn, dtype))
        if ent is not None and ent[1] >= B:
            _lib.check(_lib.lib.mgadmm_solver_set_params(ent[0], C.byref(p)))
            return ent[0], p
        if ent is not None:
            _lib.lib.mgadmm_solver_destroy(ent[0])
        h = C.c_void_p()
        _lib.check(_lib.lib.mgadmm_solver_create(gph.handle, C.byref(p), int(B), C.byref(h)))
        self._solvers[(Cn, dtype)] = [h, int(B)]
        return h, p

    def _dtype_for(self, t):
        if self.compute_dtype == 'match':
            if t.dtype not in _TORCH2MG:
                raise TypeError(f"unsupported signal dtype {t.dtype}")
            return t.dtype
        return self.compute_dtype

    def _dev_tensor(self, t, dtype, name, time_steps=None):
        if t.dim() != 4:
            raise ValueError(f"{name} must have 4 dims (B, T, N, C), got {tuple(t.shape)}")
        if t.shape[2] != self.n_nodes:
            raise ValueError(f"{name} has {t.shape[2]} nodes, graph has {self.n_nodes}")
        if time_steps is not None and t.shape[1] != time_steps:
            raise ValueError(f"{name} has {t.shape[1]} time steps, expected {time_steps}")
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _run_tensor_op(self, fn_name, x, *extra, pre=()):
        """Common driver of the tensor -> tensor entry points: (B,T,N,C) in, same shape/dtype/device out."""
        dt = self._dtype_for(x)
        xd = self._dev_tensor(x, dt, "x", self.T)
        B, _, _, Cn = xd.shape
        h, _ = self._solver(Cn, dt, B)
        ex = [self._dev_tensor(e, dt, "operand", self.T) if e is not None else None for e in extra]
        out = torch.empty_like(xd)
        fn = getattr(_lib.lib, fn_name)
        _lib.check(fn(h, *pre, _ptr(xd), *[_ptr(e) for e in ex], _ptr(out), B, _stream_ptr(xd.device)))
        return out.to(device=x.device, dtype=x.dtype)

    # ------------------------------------------------------------------ operators (ADMM.py:138-228)
    def apply_op_Lu(self, x):
        return self._run_tensor_op("mgadmm_apply", x, pre=(_lib.OP_LU,))

    def apply_op_Ldr(self, x):
        return self._run_tensor_op("mgadmm_apply", x, pre=(_lib.OP_LDR,))

    def apply_op_Ldr_T(self, x):
        return self._run_tensor_op("mgadmm_apply", x, pre=(_lib.OP_LDRT,))

    def apply_op_cLdr(self, x):
        return self._run_tensor_op("mgadmm_apply", x, pre=(_lib.OP_CLDR,))

    def apply_op_Ln(self, x):
        """Undirected temporal Laplacian (ADMM.py:248-288; unreachable from the reference's own solver: the 'UT' branch
        of LHS_zd is shadowed, ADMM.py:393-396)."""
        return self._run_tensor_op("mgadmm_apply", x, pre=(_lib.OP_LN,))

    # ------------------------------------------------------------------ left-hand sides (ADMM.py:371-399)
    def LHS_x(self, x, mask=None):
        return self._run_tensor_op("mgadmm_lhs", x, mask, pre=(_lib.LHS_X,))

    def LHS_zu(self, zu):
        return self._run_tensor_op("mgadmm_lhs", zu, None, pre=(_lib.LHS_ZU,))

    def LHS_zd(self, zd):
        if self.ablation == 'DGLR':
            print('Error: LHS_zd')      # ADMM.py:397-399
            return None
        return self._run_tensor_op("mgadmm_lhs", zd, None, pre=(_lib.LHS_ZD,))

    def phi_direct(self, x, gamma):
        """phi = soft_{mu_d1/rho}(Ldr x - gamma/rho)  (ADMM.py:401-408)."""
        return self._run_tensor_op("mgadmm_phi_direct", x, gamma)

    # ------------------------------------------------------------------ regularisers (ADMM.py:230-246)
    def DGLR(self, x):
        return (self.apply_op_Ldr(x) ** 2).sum((1, 2, 3)).mean()

    def DGTV(self, x):
        return self.apply_op_Ldr(x).abs().sum((1, 2, 3)).mean()

    def GLR(self, x):
        return (x * self.apply_op_Lu(x)).sum((1, 2, 3)).mean()

    # ------------------------------------------------------------------ CG (ADMM.py:329-368)
    def CG_solver(self, LHS_func, RHS, x0=None, **kwargs):
        """Batched CG with per-sample convergence on the GPU.  ``LHS_func`` must be one of this
        object's ``LHS_x`` / ``LHS_zu`` / ``LHS_zd`` (the operators the kernels implement).
        Returns ``(x, iters, alphas, betas)``: for B == 1 exactly the reference's shapes (int count or
        -1, float32 1-D tensors -- lists when not converged); for B > 1 a LongTensor (B,) and (K,B)
        tensors with NaN past each sample's last iteration."""
        name = getattr(LHS_func, "__name__", None)
        which = {"LHS_x": _lib.LHS_X, "LHS_zu": _lib.LHS_ZU, "LHS_zd": _lib.LHS_ZD}.get(name)
        if which is None or getattr(LHS_func, "__self__", None) is not self:
            raise NotImplementedError("CG_solver runs inside HIP kernels and only supports this object's "
                                      "LHS_x / LHS_zu / LHS_zd as LHS_func")
        mask = kwargs.get("mask")
        dt = self._dtype_for(RHS)
        rhs = self._dev_tensor(RHS, dt, "RHS", self.T)
        B, _, _, Cn = rhs.shape
        h, p = self._solver(Cn, dt, B)
        x0d = self._dev_tensor(x0, dt, "x0", self.T) if x0 is not None else None
        md = self._dev_tensor(mask, dt, "mask", self.T) if mask is not None else None
        out = torch.empty_like(rhs)
        K = p.max_cg_iter
        iters = np.zeros(B, dtype=np.int32)
        alpha = np.zeros((K, B), dtype=np.float64)
        beta = np.zeros((K, B), dtype=np.float64)
        _lib.check(_lib.lib.mgadmm_cg(h, which, _ptr(rhs), _ptr(x0d), _ptr(md), _ptr(out),
                                      iters.ctypes.data_as(C.POINTER(C.c_int32)),
                                      alpha.ctypes.data_as(C.POINTER(C.c_double)),
                                      beta.ctypes.data_as(C.POINTER(C.c_double)), B, _stream_ptr(rhs.device)))
        x = out.to(device=RHS.device, dtype=RHS.dtype)
        if B == 1:
            it = int(iters[0])
            n = it if it > 0 else K
            a = torch.tensor(alpha[:n, 0], dtype=torch.float32)
            b = torch.tensor(beta[:n, 0], dtype=torch.float32)
            if it > 0:
                return x, it, a, b
            return x, -1, list(a), list(b)
        return x, torch.from_numpy(iters.astype(np.int64)), torch.from_numpy(alpha), torch.from_numpy(beta)

    # ------------------------------------------------------------------ the ADMM loop (ADMM.py:511-648)
    def two_loops(self, y, mask=None, differential=False):
        """The reference's two-loops variant (ADMM.py:410-508): ``max_ADMM_iter`` outer phi / gamma updates around
        ``max_inner_iter`` inner (x, zu, zd, gamma_u, gamma_d) updates restarted in every outer iteration.  Like the
        reference it returns None and appends the CG counts of every inner iteration to ``CG_iter_x/zu/zd``; the final
        iterate is kept in ``self.state`` (x, zu, zd, phi, gamma, gamma_u, gamma_d)."""
        if differential:
            assert mask is None, 'differential mode does not support mask'
        dt = self._dtype_for(y)
        yd = self._dev_tensor(y, dt, "y", self.t_in if mask is None else self.T)
        B, _, N, Cn = yd.shape
        md, mask_f32 = None, 0
        if mask is not None:
            if tuple(mask.shape) != tuple(y.shape):
                raise ValueError(f"mask shape {tuple(mask.shape)} != y shape {tuple(y.shape)}")
            mask_f32 = int(mask.dtype == torch.float32)
            md = self._dev_tensor(mask, dt, "mask", self.T)
        h, p = self._solver(Cn, dt, B)
        x = torch.empty((B, self.T, N, Cn), device=yd.device, dtype=dt)
        has_phi, has_zd = self.ablation in ('None', 'DGLR'), self.ablation != 'DGLR'
        st, state = _lib.State(), {}
        for nm in ("zu", "zd", "phi", "gamma", "gamma_u", "gamma_d"):
            if nm in ("phi", "gamma") and not has_phi:
                continue
            state[nm] = torch.empty_like(x)
            setattr(st, nm, state[nm].data_ptr())
        rows = p.max_admm_iter * p.max_inner_iter
        cg_it = np.zeros((rows, 3, B), dtype=np.int32)
        hs = _lib.History()
        hs.cg_iters = cg_it.ctypes.data_as(C.POINTER(C.c_int32))
        rc = _lib.lib.mgadmm_two_loops(h, _ptr(yd), _ptr(md), mask_f32, B, _ptr(x), C.byref(st), C.byref(hs), _stream_ptr(yd.device))
        if rc == _lib.ERR_NONFINITE:
            raise AssertionError("NaN/Inf value in the two_loops iterates (reference asserts, ADMM.py:432-504): "
                                 + _lib.lib.mgadmm_last_error().decode())
        _lib.check(rc)
        for i in range(rows):
            for w, lst in ((0, self.CG_iter_x), (1, self.CG_iter_zu), (2, self.CG_iter_zd)):
                if w == 2 and not has_zd:
                    continue
                lst.append(int(cg_it[i, w, 0]) if B == 1 else torch.from_numpy(cg_it[i, w].astype(np.int64)))
        back = lambda t: t.to(device=y.device, dtype=y.dtype)
        self.state = {k2: back(v) for k2, v in state.items()}
        self.state["x"] = back(x)
        return None

    def solve(self, y, mask=None, differential=False, print_info=False, return_state=True, per_sample_history=False,
              warm_start=None):
        """Run the ADMM loop and return ``(x, (zu, zd), phi, history)``; ``history`` is a dict with the
        same lists that are also stored on the instance (p_res_list, d_res_list, ...).

        ``warm_start``: a state dict as left in ``self.state`` by a previous ``solve`` (keys x, zu, gamma_u and, as the
        ablation requires, zd, gamma_d, phi, gamma): the loop resumes from it instead of the initial guess
        (checkpoint / resume; k1 + k2 iterations in two calls equal k1 + k2 iterations in one)."""
        if differential:
            assert mask is None, 'differential mode does not support mask'   # flag has no other effect (Q3)
        dt = self._dtype_for(y)
        ts_y = self.t_in if mask is None else self.T
        yd = self._dev_tensor(y, dt, "y", ts_y)
        B, _, N, Cn = yd.shape
        md = None
        mask_f32 = 0
        if mask is not None:
            if tuple(mask.shape) != tuple(y.shape):
                raise ValueError(f"mask shape {tuple(mask.shape)} != y shape {tuple(y.shape)}")
            mask_f32 = int(mask.dtype == torch.float32)
            md = self._dev_tensor(mask, dt, "mask", self.T)
        h, p = self._solver(Cn, dt, B)
        dev = yd.device
        x = torch.empty((B, self.T, N, Cn), device=dev, dtype=dt)
        has_phi = self.ablation in ('None', 'DGLR')
        has_zd = self.ablation != 'DGLR'
        st = _lib.State()
        state = {}
        if return_state:
            for nm in ("zu", "zd", "phi", "gamma", "gamma_u", "gamma_d"):
                if (nm in ("phi", "gamma") and not has_phi):
                    continue
                state[nm] = torch.empty_like(x)
                setattr(st, nm, state[nm].data_ptr())
        K, I = p.max_cg_iter, p.max_admm_iter
        metrics = np.zeros((I, _lib.NMETRIC), dtype=np.float64)
        dxps = np.zeros((I, self.T), dtype=np.float64)
        cg_it = np.zeros((I, 3, B), dtype=np.int32)
        hs = _lib.History()
        hs.metrics = metrics.ctypes.data_as(C.POINTER(C.c_double))
        hs.delta_x_per_step = dxps.ctypes.data_as(C.POINTER(C.c_double))
        hs.cg_iters = cg_it.ctypes.data_as(C.POINTER(C.c_int32))
        mps = None
        if per_sample_history:
            mps = np.zeros((I, _lib.NMETRIC, B), dtype=np.float64)
            hs.metrics_per_sample = mps.ctypes.data_as(C.POINTER(C.c_double))
        al = be = None
        if p.record_cg_coeffs:
            al = np.full((I, 3, K, B), np.nan, dtype=np.float64)
            be = np.full((I, 3, K, B), np.nan, dtype=np.float64)
            hs.cg_alpha = al.ctypes.data_as(C.POINTER(C.c_double))
            hs.cg_beta = be.ctypes.data_as(C.POINTER(C.c_double))
        if warm_start is None:
            rc = _lib.lib.mgadmm_solve(h, _ptr(yd), _ptr(md), mask_f32, B, _ptr(x), C.byref(st), C.byref(hs),
                                       _stream_ptr(dev))
        else:
            need = ["x", "zu", "gamma_u"] + (["zd", "gamma_d"] if has_zd else []) + (["phi", "gamma"] if has_phi else [])
            missing = [k2 for k2 in need if warm_start.get(k2) is None]
            if missing:
                raise ValueError(f"warm_start misses {missing} (ablation {self.ablation!r})")
            win = {k2: self._dev_tensor(warm_start[k2], dt, "warm_start." + k2, self.T) for k2 in need}
            if any(tuple(v.shape) != tuple(x.shape) for v in win.values()):
                raise ValueError("warm_start tensors must have the shape of x (B, T, N, C)")
            sin = _lib.State()
            for nm in need[1:]:
                setattr(sin, nm, win[nm].data_ptr())
            rc = _lib.lib.mgadmm_solve_from(h, _ptr(yd), _ptr(md), mask_f32, B, _ptr(win["x"]), C.byref(sin), _ptr(x),
                                            C.byref(st), C.byref(hs), _stream_ptr(dev))
        n = hs.n_iters
        if rc == _lib.ERR_NONFINITE:
            # like the reference at its asserts (ADMM.py:534-606), the history of the iterations that ran is available
            self._fill_history(metrics[:n], dxps[:n], cg_it[:n], al, be, B, has_phi, has_zd, False)
            bad = ""
            if mps is not None:
                self.metrics_per_sample = mps[:n]
                idx = np.nonzero(~np.isfinite(mps[:n]).all(axis=(0, 1)))[0]
                self.nonfinite_samples = idx
                bad = f"; non-finite samples: {idx[:16].tolist()}{' ...' if idx.size > 16 else ''}"
            raise AssertionError("NaN/Inf value in the ADMM iterates (reference asserts, ADMM.py:534-606): "
                                 + _lib.lib.mgadmm_last_error().decode() + bad)
        _lib.check(rc)
        self._fill_history(metrics[:n], dxps[:n], cg_it[:n], al, be, B, has_phi, has_zd, print_info)
        if mps is not None:
            self.metrics_per_sample = mps[:n]
        back = lambda t: t.to(device=y.device, dtype=y.dtype)
        xo = back(x)
        zu = back(state["zu"]) if "zu" in state else None
        zd = back(state["zd"]) if "zd" in state else None
        phi = back(state["phi"]) if "phi" in state else None
        self.state = {k2: back(v) for k2, v in state.items()}
        self.state["x"] = xo
        return xo, (zu, zd), phi, self.history()

    def combined_loop(self, y, mask=None, differential=False, print_info=True):
        """``y`` (B, t_in, N, C) [or (B, T, N, C) with ``mask``] -> ``x`` (B, T, N, C), dtype/device of y.
        History attributes are filled like the reference's (ADMM.py:612-643)."""
        return self.solve(y, mask=mask, differential=differential, print_info=print_info, return_state=False)[0]

    def history(self):
        keys = ("p_res_list", "d_res_list", "x_shift_list", "delta_x_per_step", "GLR_list", "DGTV_list", "DGLR_list",
                "recover_list", "CG_iter_x", "CG_iter_zu", "CG_iter_zd", "alpha_x", "beta_x", "alpha_zu", "beta_zu",
                "alpha_zd", "beta_zd", "res_name")
        return {k2: getattr(self, k2) for k2 in keys}

    def _fill_history(self, metrics, dxps, cg_it, al, be, B, has_phi, has_zd, print_info):
        L = _lib
        for i in range(metrics.shape[0]):
            m = metrics[i]
            pri, dual = [float(m[L.M_PRI_ZU])], [float(m[L.M_DUAL_ZU])]
            self.x_shift_list.append(float(m[L.M_XSHIFT]))
            self.delta_x_per_step.append(torch.tensor(dxps[i]))
            self.GLR_list.append(torch.tensor(m[L.M_GLR]))
            self.recover_list.append(float(m[L.M_RECOVER]))
            if has_phi:
                pri.append(float(m[L.M_PRI_PHI])); dual.append(float(m[L.M_DUAL_PHI]))
                self.DGTV_list.append(torch.tensor(m[L.M_DGTV]))
            if has_zd:
                pri.append(float(m[L.M_PRI_ZD])); dual.append(float(m[L.M_DUAL_ZD]))
                self.DGLR_list.append(torch.tensor(m[L.M_DGLR]))
            self.p_res_list.append(pri)
            self.d_res_list.append(dual)
            its = []
            for w, lst in ((0, self.CG_iter_x), (1, self.CG_iter_zu), (2, self.CG_iter_zd)):
                if w == 2 and not has_zd:
                    its.append(None)
                    continue
                v = int(cg_it[i, w, 0]) if B == 1 else torch.from_numpy(cg_it[i, w].astype(np.int64))
                lst.append(v)
                its.append(v)
            if al is not None:
                for w, la, lb in ((0, self.alpha_x, self.beta_x), (1, self.alpha_zu, self.beta_zu),
                                  (2, self.alpha_zd, self.beta_zd)):
                    if w == 2 and not has_zd:
                        continue
                    if B == 1:
                        kk = int(cg_it[i, w, 0])
                        kk = kk if kk > 0 else al.shape[2]
                        la.append(torch.tensor(al[i, w, :kk, 0], dtype=torch.float32))
                        lb.append(torch.tensor(be[i, w, :kk, 0], dtype=torch.float32))
                    else:
                        la.append(torch.from_numpy(al[i, w]))
                        lb.append(torch.from_numpy(be[i, w]))
            if print_info:
                fmt = lambda v: str(v) if not torch.is_tensor(v) else f"{int(v.min())}..{int(v.max())}"
                zd_s = fmt(its[2]) if its[2] is not None else "n/a"
                print(f'ADMM iters {i}: x_CG_iters {fmt(its[0])}, zu_CG_iters {fmt(its[1])}, zd_CG_iters {zd_s}, '
                      f'pri_err = [{", ".join([f"{e:.4g}" for e in pri])}], '
                      f'dual_err = [{", ".join([f"{e:.4g}" for e in dual])}]')

    # ------------------------------------------------------------------ profiling hooks (bench.py)
    def prof_begin(self, Cn=1, dtype=torch.float32):
        h = self._solvers[(Cn, dtype)][0]
        _lib.check(_lib.lib.mgadmm_prof_begin(h))

    def prof_end(self, Cn=1, dtype=torch.float32):
        h = self._solvers[(Cn, dtype)][0]
        cnt = (C.c_int64 * _lib.NPROF)()
        ms = (C.c_double * _lib.NPROF)()
        by = (C.c_double * _lib.NPROF)()
        _lib.check(_lib.lib.mgadmm_prof_end(h, cnt, ms, by))
        return [dict(count=int(cnt[i]), ms=float(ms[i]), bytes=float(by[i])) for i in range(_lib.NPROF)]

    def workspace_bytes(self, Cn=1, dtype=torch.float32):
        return int(_lib.lib.mgadmm_solver_workspace_bytes(self._solvers[(Cn, dtype)][0]))

    # ------------------------------------------------------------------ plotting (ADMM.py:650-761)
    def _plot(self, curves, labels, title, descriptions, save_path, log_y, x0=0):
        import matplotlib.pyplot as plt
        plt.figure()
        plt.grid()
        curves = torch.as_tensor(curves)
        plt.plot(torch.arange(x0, x0 + curves.shape[0], 1), curves)
        if labels:
            plt.legend(labels)
        plt.title(f'{title} ({descriptions})' if descriptions is not None else title)
        plt.xlabel('ADMM iterations')
        if log_y:
            plt.yscale('log')
        plt.show()
        if save_path is not None:
            plt.savefig(save_path)
        plt.close()

    def plot_residual(self, descriptions=None, save_path=None, log_y=False):
        res = torch.cat((torch.Tensor(self.p_res_list), torch.Tensor(self.d_res_list),
                         torch.Tensor(self.x_shift_list).reshape(-1, 1)), 1)
        legend = ['pri_' + s for s in self.res_name] + ['dual_' + s for s in self.res_name] + ['dual_x']
        self._plot(res, legend, 'Residuals in ADMM', descriptions, save_path, log_y)

    def plot_x_per_step(self, save_path=None, show_list=None, start_iters=0, descriptions=None, log_y=False):
        dxps = torch.stack(self.delta_x_per_step, dim=0)
        show_list = list(range(self.T)) if show_list is None else show_list
        self._plot(dxps[start_iters:, show_list], [f'dx_{j}' for j in show_list], 'Delta_x for each time step',
                   descriptions, save_path, log_y, x0=start_iters)

    def plot_CG_params(self, descriptions=None, save_path=None, discriptions=None, log_y=False):
        res = torch.cat((torch.Tensor(self.p_res_list), torch.Tensor(self.d_res_list)), 1)
        legend = ['alpha_' + s for s in self.res_name] + ['beta_' + s for s in self.res_name]
        self._plot(res, legend, 'CGD params in ADMM', descriptions, save_path, True)

    def plot_regularization_terms(self, save_path=None, descriptions=None, log_y=False):
        cols, labels = [torch.Tensor(self.GLR_list)], ['GLR']
        if self.ablation != 'DGLR':
            cols.append(torch.Tensor(self.DGLR_list)); labels.append('DGLR')
        if self.ablation in ['DGLR', 'None']:
            cols.append(torch.Tensor(self.DGTV_list)); labels.append('DGTV')
        cols.append(torch.Tensor(self.recover_list)); labels.append('||Hx - y||')
        self._plot(torch.stack(cols, 1), labels, 'Regularization terms in ADMM', descriptions, save_path, log_y)


# ---------------------------------------------------------------------- module functions (ADMM.py:766-811)
def _standalone(y, T, t_in):
    """A throw-away solver on an edgeless graph: the initial guesses do not touch the graph."""
    N = y.shape[2]
    gi = {'n_nodes': N}
    info = dict(rho=1.0, rho_u=1.0, rho_d=1.0, mu_u=1.0, mu_d1=1.0, mu_d2=1.0)
    cl = torch.stack([torch.arange(N), torch.full((N,), -1, dtype=torch.int64)], 1)
    dl = torch.tensor([0.0, float('inf')]).repeat(N, 1)
    return ADMM_algorithm(gi, info, use_kNN=True, k=1, u_sigma=1.0, d_sigma=1.0, t_in=t_in, T=T, tables=(cl, dl),
                          compute_dtype='match', reorder=False)


def initial_guess(y, t_in, T):
    """Least-squares line through the observed steps, extrapolated to T (ADMM.py:766-781)."""
    blk = _standalone(y, T, t_in)
    dt = blk._dtype_for(y)
    yd = blk._dev_tensor(y, dt, "y", t_in)
    B, _, N, Cn = yd.shape
    h, _ = blk._solver(Cn, dt, B)
    x = torch.empty((B, T, N, Cn), device=yd.device, dtype=dt)
    _lib.check(_lib.lib.mgadmm_initial_guess(h, _ptr(yd), _ptr(x), B, _stream_ptr(yd.device)))
    out = x.to(device=y.device, dtype=y.dtype)
    blk.close()
    return out


def initial_interpolation(y, mask):
    """Masked regression fill-in (ADMM.py:783-811)."""
    T = y.shape[1]
    blk = _standalone(y, T, min(T, 2))
    dt = blk._dtype_for(y)
    yd = blk._dev_tensor(y, dt, "y", T)
    md = blk._dev_tensor(mask, dt, "mask", T)
    B, _, N, Cn = yd.shape
    h, _ = blk._solver(Cn, dt, B)
    x = torch.empty_like(yd)
    _lib.check(_lib.lib.mgadmm_initial_interpolation(h, _ptr(yd), _ptr(md), int(mask.dtype == torch.float32), _ptr(x), B,
                                                     _stream_ptr(yd.device)))
    out = x.to(device=y.device, dtype=y.dtype)
    blk.close()
    return out
