"""CPU check of the LDS bank model and the entry-order search of the LDS-resident kernel's gathers
(mixed-graph-admm_amd/csrc/lds_banks.h, plain C++): tests/cpu/lds_banks_check.cpp replays the read stream of k_admm_lds on a
graph, runs the search and checks that every row keeps its multiset of columns, that `src` is the permutation applied, that
the conflict count did not grow and that it equals a fresh replay.  Built with AddressSanitizer + UBSan."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("banks") / "lds_banks_check")
    subprocess.check_call([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(PKG, "csrc"), os.path.join(ROOT, "tests", "cpu", "lds_banks_check.cpp"), "-o", exe])
    return exe


def _graph(tmp_path, n, k, seed, chords):
    """kNN-like table: self + k neighbours per row, mostly path neighbours, `chords` random far columns."""
    from export_lds_graph import export
    rng = np.random.default_rng(seed)
    cl = np.zeros((n, k + 1), dtype=np.int64)
    for i in range(n):
        near = [j for j in (i - 1, i + 1, i - 2, i + 2, i - 3, i + 3, i - 4, i + 4) if 0 <= j < n][:k]
        while len(near) < k:
            near.append(int(rng.integers(n)))
        cl[i] = [i] + near
    for _ in range(chords):
        i, j = rng.integers(n, size=2)
        cl[i, 1 + rng.integers(k)] = j
    path = str(tmp_path / f"g{n}_{k}_{seed}.graph")
    export(cl, path)
    return path


@pytest.mark.parametrize("n,k,chords,mode", [(307, 4, 60, "targeted"), (170, 4, 30, "targeted"), (307, 4, 60, "random"),
                                              (100, 3, 20, "targeted"), (64, 6, 10, "rows")])
def test_bank_search_keeps_the_tables_and_lowers_the_conflicts(checker, tmp_path, n, k, chords, mode):
    path = _graph(tmp_path, n, k, seed=n + k, chords=chords)
    args = {"targeted": ["400", "0", "1", "1"], "random": ["4000", "0", "1", "0"], "rows": ["1500", "1", "0", "0"]}[mode]
    out = subprocess.run([checker, path] + args, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK"), out.stdout
    m = re.search(r"weighted conflicts ([\d.]+) -> ([\d.]+)", out.stdout)
    before, after = float(m.group(1)), float(m.group(2))
    assert after <= before
    if mode == "targeted" and n >= 170:
        assert after < 0.6 * before, out.stdout          # the search is worth its host time
