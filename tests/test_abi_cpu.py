"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mgadmm.h declares; error paths that need no GPU compute."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "mgadmm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mgadmm_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    fns = header_functions()
    for must in ("mgadmm_graph_create", "mgadmm_solver_create", "mgadmm_apply", "mgadmm_lhs", "mgadmm_cg",
                 "mgadmm_solve", "mgadmm_phi_direct", "mgadmm_initial_guess", "mgadmm_initial_interpolation"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from mgadmm import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), f"libmgadmm.so does not export {name}"
        assert name in _lib.SYMBOLS, f"ctypes binding misses {name}"
    assert set(_lib.SYMBOLS) == set(header_functions())
    assert "gfx950" in _lib.version()


def test_struct_layouts_match_header_field_order():
    from mgadmm import _lib
    txt = open(os.path.join(ROOT, "include", "mgadmm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)

    def fields(struct_name):
        body = re.search(r"typedef struct \{([^{}]*)\} " + struct_name + ";", txt).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())[0])
        return names

    assert fields("mgadmm_graph_desc") == [f[0] for f in _lib.GraphDesc._fields_]
    assert fields("mgadmm_params") == [f[0] for f in _lib.Params._fields_]
    assert fields("mgadmm_history") == [f[0] for f in _lib.History._fields_]
    assert fields("mgadmm_state") == [f[0] for f in _lib.State._fields_]


def test_invalid_arguments_are_reported_without_a_gpu():
    from mgadmm import _lib
    h = ctypes.c_void_p()
    rc = _lib.lib.mgadmm_graph_create(None, ctypes.byref(h))
    assert rc == _lib.ERR_INVALID
    assert b"null" in _lib.lib.mgadmm_last_error()
    d = _lib.GraphDesc()
    d.n_nodes, d.T = 0, 24
    assert _lib.lib.mgadmm_graph_create(ctypes.byref(d), ctypes.byref(h)) == _lib.ERR_INVALID
    with pytest.raises(_lib.MgadmmError):
        _lib.check(_lib.lib.mgadmm_apply(None, 0, None, None, 1, None))


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mgadmm.ADMM import ADMM_algorithm
    import torch as th
    cl = th.tensor([[0, 1], [1, 0]])
    blk = ADMM_algorithm({"n_nodes": 2}, dict(rho=1, rho_u=1, rho_d=1, mu_u=1, mu_d1=1, mu_d2=1), use_kNN=True,
                         u_sigma=1.0, d_sigma=1.0, tables=(cl, th.tensor([[0.0, 1.0], [0.0, 1.0]])))
    with pytest.raises(RuntimeError, match="no CPU path"):
        blk.combined_loop(th.zeros(1, 12, 2, 1), print_info=False)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mixed-graph-admm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("test oracle", ""), f"{f} mentions the oracle"
