"""Writes the three CSR structures of a bench workload (W_u, W_d, W_d^T: columns only, table order) in the text format of
tests/cpu/lds_banks_check.cpp.      python tools/export_lds_graph.py cfg2 /tmp/cfg2.graph"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import numpy as np


def csr_text(rows, stream, weight):
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    col = [c for r in rows for c in r]
    return f"{stream} {weight} {len(col)}\n" + " ".join(map(str, rp)) + "\n" + " ".join(map(str, col)) + "\n"


def export(cl, path, T=24, tpg=8, nlead=4):
    n = cl.shape[0]
    ts = (T + 3) // 4 * 4
    if (ts // 4) % 2 == 0:
        ts += 4
    wu = [[int(c) for c in cl[i, 1:] if c >= 0] for i in range(n)]
    wd = [[int(c) for c in cl[i] if c >= 0] for i in range(n)]
    wt = [[] for _ in range(n)]
    for i in range(n):
        for c in wd[i]:
            wt[c].append(i)
    uni = all(len(r) == 4 for r in wu) and all(len(r) == 5 for r in wd)
    with open(path, "w") as f:
        f.write(f"{n} {T // tpg} {tpg} {ts} {nlead}\n")
        f.write(csr_text(wu, 0 if uni else 2, 11.6))
        f.write(csr_text(wd, 0 if uni else 2, 35.1))
        f.write(csr_text(wt, 1, 36.1))


if __name__ == "__main__":
    import bench
    n, B, cl, dl, info, desc = bench.build_problem(sys.argv[1])
    export(cl.numpy(), sys.argv[2])
