"""CPU check of the fused cLdr kernel's tile tables (mixed-graph-admm_amd/csrc/cldr_tiles.h, plain C++): the checker
tests/cpu/cldr_tiles_check.cpp replays the kernel's dataflow on the host from the tables and compares it with
Ldr^T(Ldr x) taken directly from the CSR matrices (operator definitions of reference ADMM.py:150-228)."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, ROOT


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("cldr") / "cldr_tiles_check")
    # AddressSanitizer + UBSan build (sanitizers on the CPU-buildable host code): an out-of-range
    # table index in the builder or in the replay aborts the run
    subprocess.check_call([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(PKG, "csrc"),
                           os.path.join(ROOT, "tests", "cpu", "cldr_tiles_check.cpp"), "-o", exe])
    return exe


# n, T, Rcap, C1cap, C2cap, GD, GT, cluster
@pytest.mark.parametrize("args", [
    (3000, 6, 64, 88, 120, 8, 12, 64),     # float32 production geometry
    (3000, 4, 32, 56, 80, 8, 12, 64),      # float64 geometry (clusters are halved to fit)
    (1000, 3, 64, 88, 120, 8, 12, 37),     # tiles not aligned to the caps
    (500, 2, 8, 24, 48, 8, 12, 8),         # tiny tiles, T = 2 (first step is also the one before the last)
    (777, 5, 64, 40, 60, 8, 12, 64),       # tight halo caps force repeated halving
    (64, 7, 64, 88, 120, 8, 12, 64),       # one tile holds the whole graph
])
def test_tile_tables_reproduce_cldr(checker, args):
    out = subprocess.run([checker] + [str(a) for a in args], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout
