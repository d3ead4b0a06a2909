"""Where one launch of k_admm_lds spends its time, phase by phase (cfg2 workload).

Needs a diagnostic build of the library in which the per-sample metric slots receive the 100 MHz clock at the phase
boundaries instead of the metrics:

    make -C mixed-graph-admm_amd/csrc OUT=$PWD/ko/lib_clock.so OBJDIR=$PWD/ko/build_clock EXTRA=-DMGADMM_PHASE_CLOCK
    MGADMM_LIB=$PWD/ko/lib_clock.so python tools/lds_phase_clock.py [iterations]      (MGADMM_LDS_CHUNK, MGADMM_LDS_TPG as usual)

Prints, per ADMM iteration, the mean over the samples of every phase in microseconds and the share of the workgroup's
lifetime, plus the gap between the end of one workgroup and the start of the next one on the same CU slot (estimated
from the sorted start / end times).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

NAMES = ["operands of RHS_x (HBM at the first trip of a launch, registers / LDS slot later)", "RHS_x (first trip of a cold start: + phi = Ldr x0)",
         "x solve", "store x, x metrics, operands of the zu solve", "zu solve", "zu / gamma_u update, operands of the zd solve",
         "zd solve", "zd / gamma_d update, operands of the prox", "phi prox + Ldr / Lu diagnostics", "metric totals"]


def _mean(v):
    return float(torch.as_tensor(v).double().mean())


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    n, B, cl, dl, info, _ = bench.build_problem("cfg2")
    B = int(os.environ.get("CLOCK_B", B))
    blk = bench.make_solver(n, cl, dl, info, dev)
    blk.max_ADMM_iter = iters
    y = bench.synth_y(n, B, 12, 0, 0, dev)
    try:
        blk.solve(y, per_sample_history=True)
    except AssertionError:
        pass                                  # the "metrics" are clock values; nothing checks them, but stay safe
    m = np.asarray(blk.metrics_per_sample)    # [iters][11][B]  clock ticks of 10 ns
    for it in range(m.shape[0]):
        t = m[it] * 0.01                      # microseconds
        d = np.diff(t, axis=0)                # [10][B]
        life = t[10] - t[0]
        print(f"ADMM iteration {it}: workgroup lifetime mean {life.mean():.1f} us (min {life.min():.1f}, max {life.max():.1f}); "
              f"launch span {t[10].max() - t[0].min():.1f} us; CG iterations x/zu/zd "
              f"{_mean(blk.CG_iter_x[it]):.2f} / {_mean(blk.CG_iter_zu[it]):.2f} / {_mean(blk.CG_iter_zd[it]):.2f}")
        for k, nm in enumerate(NAMES):
            print(f"    {nm:88s} {d[k].mean():8.2f} us  {100 * d[k].mean() / life.mean():5.1f} %")
        # hand-over gap: k-th earliest end vs (256 + k)-th earliest start (one workgroup per CU, 256 CUs)
        st, en = np.sort(t[0]), np.sort(t[10])
        ncu = 256
        if B > ncu:
            gap = st[ncu:] - en[:B - ncu]
            print(f"    gap between a workgroup's end and its successor's first instruction: mean {gap.mean():.2f} us")
        print(f"    busy share of the launch span: {life.sum() / ncu / (t[10].max() - t[0].min()):.3f}")


if __name__ == "__main__":
    main()
