"""GPU experiment: one cfg2 batch as TWO half-batch solves running concurrently on two streams (two solver instances, two host
threads; ctypes releases the GIL), against one whole-batch solve.  Samples are independent: same iterates.  The question is
whether the drain of one stream's k_admm_lds launch is filled by the other stream's workgroups."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mixed-graph-admm_amd"))
import torch
import bench

dev = torch.device("cuda", 0)
n, B, cl, dl, info, desc = bench.build_problem("cfg2")
K = 20
y = bench.synth_y(n, B, 12, 1, 0, dev)
whole = bench.make_solver(n, cl, dl, info, dev)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
halves = [bench.make_solver(n, cl, dl, info, dev) for _ in range(parts)]
streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
bounds = [(i * B // parts, (i + 1) * B // parts) for i in range(parts)]


def solve_whole():
    whole.max_ADMM_iter = K; whole._reset_history()
    return whole.combined_loop(y, print_info=False)


def solve_split():
    out = [None] * parts
    def work(i):
        with torch.cuda.stream(streams[i]):
            blk = halves[i]; blk.max_ADMM_iter = K; blk._reset_history()
            out[i] = blk.combined_loop(y[bounds[i][0]:bounds[i][1]], print_info=False)
    th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
    for t in th: t.start()
    for t in th: t.join()
    return torch.cat(out, 0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, x

for r in range(2):
    tw, xw = timed(solve_whole)
    ts, xs = timed(solve_split)
    print(f"whole {B*K/tw:10.0f} it/s ({tw*1e3:.2f} ms)   {parts} concurrent parts {B*K/ts:10.0f} it/s ({ts*1e3:.2f} ms)   bitwise equal: {bool(torch.equal(xw, xs))}", flush=True)
