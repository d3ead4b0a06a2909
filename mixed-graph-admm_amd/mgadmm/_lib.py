"""ctypes binding of libmgadmm.so (C ABI declared in include/mgadmm.h).

The product path has no CPU fallback: importing this module fails loudly when the HIP library has
not been built (run ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C mixed-graph-admm_amd/csrc``).
"""
import ctypes as C
import os

# torch first: it bundles its own libamdhip64.so.7; loading it before libmgadmm.so makes both share ONE
# HIP runtime (device pointers, streams and events are exchanged between them).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MGADMM_LIB selects another build of the same library (kernel experiments with different compiler flags)
LIB_PATH = os.environ.get("MGADMM_LIB") or os.path.join(_HERE, "libmgadmm.so")

OK, ERR_INVALID, ERR_HIP, ERR_NONFINITE, ERR_UNSUPPORTED, ERR_NOMEM = 0, -1, -2, -3, -4, -5
F32, F64 = 0, 1
TEMPORAL_SPATIAL, TEMPORAL_BAND = 0, 1
OP_LU, OP_LDR, OP_LDRT, OP_CLDR, OP_LN = 0, 1, 2, 3, 4
CG_PER_SAMPLE, CG_BATCH_MAX = 0, 1
LHS_X, LHS_ZU, LHS_ZD = 0, 1, 2
ABLATIONS = {"None": 0, "DGTV": 1, "DGLR": 2, "UT": 3}
PATH_AUTO, PATH_STREAM, PATH_LDS = 0, 1, 2
NMETRIC = 11
(M_XSHIFT, M_PRI_ZU, M_DUAL_ZU, M_PRI_PHI, M_DUAL_PHI, M_PRI_ZD, M_DUAL_ZD, M_GLR, M_DGTV, M_DGLR,
 M_RECOVER) = range(11)
NPROF = 4
(Q_LDS_OK, Q_LDS_TPG, Q_LDS_THREADS, Q_LDS_BYTES, Q_LDS_ROW_STRIDE, Q_NNZ_U, Q_NNZ_D, Q_NNZ_DT, Q_TILE_ROWS, Q_LDS_UNIFORM,
 Q_LDS_TAIL_PAIRS, Q_LDS_LEAD, Q_LDS_SLOTS, Q_LDS_CHUNK, Q_LDS_ROWS, Q_CLDR_SLOTS) = range(16)

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


class GraphDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("T", C.c_int32), ("temporal_mode", C.c_int32),
        ("u_rowptr", _i32p), ("u_col", _i32p), ("u_val", _f32p),
        ("d_rowptr", _i32p), ("d_col", _i32p), ("d_val", _f32p),
        ("transpose_by_gather", C.c_int32), ("q1_identity_t0", C.c_int32),
        ("skip", C.c_int32), ("band_w", _f32p),
        ("reorder", C.c_int32), ("device", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("rho", C.c_double), ("rho_u", C.c_double), ("rho_d", C.c_double),
        ("mu_u", C.c_double), ("mu_d1", C.c_double), ("mu_d2", C.c_double),
        ("t_in", C.c_int32), ("ablation", C.c_int32),
        ("cg_tol", C.c_double), ("max_cg_iter", C.c_int32),
        ("admm_tol", C.c_double), ("max_admm_iter", C.c_int32),
        ("dtype", C.c_int32), ("check_stop", C.c_int32), ("path", C.c_int32),
        ("record_cg_coeffs", C.c_int32), ("cg_convergence", C.c_int32), ("max_inner_iter", C.c_int32),
    ]


class History(C.Structure):
    _fields_ = [
        ("n_iters", C.c_int32),
        ("metrics", _f64p), ("delta_x_per_step", _f64p), ("cg_iters", _i32p),
        ("metrics_per_sample", _f64p), ("cg_alpha", _f64p), ("cg_beta", _f64p),
    ]


class State(C.Structure):
    _fields_ = [("zu", C.c_void_p), ("zd", C.c_void_p), ("phi", C.c_void_p), ("gamma", C.c_void_p),
                ("gamma_u", C.c_void_p), ("gamma_d", C.c_void_p)]


# every symbol include/mgadmm.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "mgadmm_version": (C.c_char_p, []),
    "mgadmm_last_error": (C.c_char_p, []),
    "mgadmm_graph_create": (C.c_int, [C.POINTER(GraphDesc), C.POINTER(_vp)]),
    "mgadmm_graph_destroy": (C.c_int, [_vp]),
    "mgadmm_graph_transpose_nnz": (C.c_int, [_vp, _i32p]),
    "mgadmm_graph_get_transpose": (C.c_int, [_vp, _i32p, _i32p, _f32p]),
    "mgadmm_graph_get_perm": (C.c_int, [_vp, _i32p]),
    "mgadmm_solver_create": (C.c_int, [_vp, C.POINTER(Params), C.c_int32, C.POINTER(_vp)]),
    "mgadmm_solver_destroy": (C.c_int, [_vp]),
    "mgadmm_solver_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "mgadmm_solver_workspace_bytes": (C.c_int64, [_vp]),
    "mgadmm_solver_path": (C.c_int, [_vp, C.c_int32]),
    "mgadmm_solver_query": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64)]),
    "mgadmm_apply": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp]),
    "mgadmm_lhs": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp]),
    "mgadmm_phi_direct": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "mgadmm_initial_guess": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp]),
    "mgadmm_initial_interpolation": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp]),
    "mgadmm_cg": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _i32p, _f64p, _f64p, C.c_int32, _vp]),
    "mgadmm_solve": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.POINTER(State), C.POINTER(History), _vp]),
    "mgadmm_solve_from": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.POINTER(State), _vp, C.POINTER(State), C.POINTER(History), _vp]),
    "mgadmm_two_loops": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp, C.POINTER(State), C.POINTER(History), _vp]),
    "mgadmm_knn_graph": (C.c_int, [C.c_int32, C.c_int64, C.POINTER(C.c_int64), _f64p, C.c_int32, _i32p, _f32p, C.c_int32]),
    "mgadmm_weight_tables": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int64), _f32p, C.c_double, C.c_double, C.c_int32,
                                       _f32p, _f32p, _f64p, C.c_int32]),
    "mgadmm_series_stats": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "mgadmm_series_affine": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp]),
    "mgadmm_gather_windows": (C.c_int, [_vp, C.c_int64, C.c_int32, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "mgadmm_prof_begin": (C.c_int, [_vp]),
    "mgadmm_prof_end": (C.c_int, [_vp, C.POINTER(C.c_int64), _f64p, _f64p]),
}


class MgadmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmgadmm error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. mgadmm has no CPU fallback; "
            "build it with `make -C mixed-graph-admm_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    if rc != OK:
        raise MgadmmError(rc, lib.mgadmm_last_error().decode("utf-8", "replace"))
    return rc


def version():
    return lib.mgadmm_version().decode()


def query(handle, what):
    out = C.c_int64()
    check(lib.mgadmm_solver_query(handle, what, C.byref(out)))
    return int(out.value)
