"""CPU-baseline worker of bench.py  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (never part of the product path).

One process = one host core.  It runs the oracle (oracle/admm_oracle.py, the CPU restatement of reference
ADMM.py:511-648) over its share of the sample windows, either

  mode "loop"   one window at a time, B = 1 -- the reference's own semantics (it cannot run B > 1, ADMM.py:362), or
  mode "batch"  all its windows as one vectorised batch (per-sample CG convergence),

and reports the wall time of the solve alone.  Protocol (so that all workers start together and import / set-up time
is not measured): load inputs, build the oracle, print "READY", wait for a line on stdin, solve, print one JSON line.

    python oracle/cpu_bench_worker.py <inputs.npz> <lo> <hi> <mode> <n_iters>
"""
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402

from oracle import admm_oracle as orc  # noqa: E402


def main():
    path, lo, hi, mode, n_iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    d = np.load(path, allow_pickle=False)
    info = {k: float(d[k]) for k in ("rho", "rho_u", "rho_d", "mu_u", "mu_d1", "mu_d2")}
    o = orc.OracleADMM(d["cl"], d["u_ew"], d["d_ew"], info, mode="knn", t_in=int(d["t_in"]), T=int(d["T"]))
    y = d["y"][lo:hi]
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    if mode == "loop":
        for b in range(y.shape[0]):
            o.combined_loop(y[b:b + 1], n_iters=n_iters)
    else:
        o.combined_loop(y, n_iters=n_iters)
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "samples": int(y.shape[0]), "iters": n_iters}), flush=True)


if __name__ == "__main__":
    main()
