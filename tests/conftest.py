import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mixed-graph-admm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g4_meta():
    return load_golden("g4_meta.npz")


@pytest.fixture(scope="session")
def g4_solves():
    return load_golden("g4_solves.npz")


def admm_info_from(meta):
    return {k: float(meta[k]) for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2")}


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _ensure_built():
    """The product package cannot be imported without libmgadmm.so.  `make` is run whenever hipcc is present (a no-op when
    the library is newer than every source; hipcc cross-compiles gfx950 without a GPU), so the tests never run against a
    binary built from older sources than the tree."""
    import shutil
    import subprocess
    so = os.path.join(PKG, "mgadmm", "libmgadmm.so")
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc):
        subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc"), f"HIPCC={hipcc}"])
    elif not os.path.exists(so):
        raise RuntimeError(f"{so} is missing and hipcc was not found: build it with `make -C mixed-graph-admm_amd/csrc`")


_ensure_built()
