#!/usr/bin/env python3
"""Benchmark of the ADMM hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one ADMM iteration (3 batched CG solves + prox + dual updates + residual history,
reference ADMM.py:546-646) over the whole batch resident in HBM.  The timed region is ONE solve of
exactly K iterations (no early stop), including its small set-up (layout pack, initial guess) and the
final unpack / gather, bracketed by barrier + torch.cuda.synchronize(); the time is the MAX over ranks.

Workloads (BASELINE.json configs; synthetic data, seeds fixed):
  cfg2 (default) PEMS04-shaped graph: N=307, 340 undirected edges, k=4, B=4096 windows per GPU, fp32
  cfg3           N=10 000 Euclidean kNN graph, B=512 per GPU, fp32     (CG-SpMV HBM roofline run)
  cfg4           N=100 000 mixed graph, B=256 per GPU (weak scaling 1..8 GPUs), fp32
Metric: ADMM sample-iterations per second = (samples over all ranks) * K / seconds.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded around every launch
of the dominant kernel (the sparse-Laplacian SpMM inside the CG solves) on the launch stream, during the
timed solve; `cpu_baseline` is the CPU oracle (a NumPy/SciPy port of the reference algorithm, 1 thread)
timed on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mixed-graph-admm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def pems_like_graph(n, n_edges, seed=0):
    """Path + random chords with PEMS-like edge lengths U(3, 600)."""
    rng = random.Random(seed)
    edges = [(i, i + 1) for i in range(n - 1)]
    have = set(edges) | {(b, a) for a, b in edges}
    while len(edges) < n_edges:
        a, b = rng.randrange(n), rng.randrange(n)
        if a == b or (a, b) in have:
            continue
        edges.append((a, b)); have.add((a, b)); have.add((b, a))
    d = [rng.uniform(3.0, 600.0) for _ in edges]
    e = np.array(edges, dtype=np.int64)
    d = np.array(d)
    u_edges = torch.from_numpy(np.concatenate([e, e[:, ::-1]], 0).copy())
    u_dist = torch.from_numpy(np.concatenate([d, d]))
    return u_edges, u_dist


def build_problem(workload):
    import mgadmm
    if workload == "cfg1":          # the reference's own CPU-runnable case (parity test, not a bench line)
        n, B = 170, 1
        ue, ud = pems_like_graph(n, 295, seed=0)
        cl, dl = mgadmm.utils.k_nearest_neighbors(n, ue, ud, 4)
        cl = cl.to(torch.int64)
        desc = "PEMS08-shaped synthetic graph N=170 (path+chords, 295 edges), kNN k=4, T=24, t_in=12"
    elif workload == "cfg2":
        n, B = 307, 4096
        ue, ud = pems_like_graph(n, 340, seed=0)
        cl, dl = mgadmm.utils.k_nearest_neighbors(n, ue, ud, 4)
        cl = cl.to(torch.int64)
        desc = "PEMS04-shaped synthetic graph N=307 (path+chords, 340 edges), kNN k=4, T=24, t_in=12"
    elif workload == "cfg3":
        n, B = 10000, 512
        pts = np.random.default_rng(0).random((n, 2))
        cl, dl = mgadmm.utils.knn_from_points(pts, 4, scale=1000.0)
        desc = "synthetic 10k-node Euclidean kNN graph (k=4), T=24, t_in=12"
    elif workload == "cfg4":
        n, B = 100000, 256
        pts = np.random.default_rng(0).random((n, 2))
        cl, dl = mgadmm.utils.knn_from_points(pts, 4, scale=3000.0)
        desc = "synthetic 100k-node mixed graph (asymmetric kNN W_d, normalised kNN W_u), T=24, t_in=12"
    else:
        raise SystemExit(f"unknown workload {workload}")
    r = math.sqrt(n / 24)
    info = dict(rho=2 * r, rho_u=3 * r, rho_d=2 * r, mu_u=1, mu_d1=2, mu_d2=1)
    return n, B, cl, dl, info, desc


def synth_y(n, B, t_in, seed, offset, device):
    """Smooth synthetic traffic: a_n + b_n sin(2 pi (t + tau_b) / 288) + 5 randn, consecutive windows."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = 50 + 350 * torch.rand(n, generator=g)
    b = 20 + 60 * torch.rand(n, generator=g)
    ph = 2 * math.pi * torch.rand(n, generator=g)
    a, b, ph = a.to(device), b.to(device), ph.to(device)
    t = torch.arange(t_in, device=device, dtype=torch.float32)[None, :, None]
    tau = (offset + torch.arange(B, device=device, dtype=torch.float32))[:, None, None]
    gd = torch.Generator(device=device).manual_seed(seed + 1 + offset)
    y = a[None, None, :] + b[None, None, :] * torch.sin(2 * math.pi * (t + tau) / 288 + ph[None, None, :])
    y = y + 5 * torch.randn(B, t_in, n, generator=gd, device=device)
    return y.unsqueeze(-1).contiguous()


def make_solver(n, cl, dl, info, device):
    import mgadmm
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl),
                                device=device, compute_dtype=torch.float32, record_cg_coeffs=False)
    blk.check_stop = False          # fixed iteration count: exactly K steps
    return blk


def cpu_baseline(n, cl, info, blk, y_dev, budget_s, steps):
    """Oracle (CPU port of the reference algorithm) on a bounded sample of the same workload."""
    from oracle import admm_oracle as orc
    o = orc.OracleADMM(cl.numpy(), blk.u_ew[0].numpy(), blk.d_ew[0].numpy(), info, mode="knn", t_in=12, T=24)
    # calibrate on a small slice, then size the timed sample to ~budget_s seconds of CPU work
    yc0 = y_dev[:8].double().cpu().numpy()
    t0 = time.perf_counter()
    o.combined_loop(yc0, n_iters=2)
    per = (time.perf_counter() - t0) / 16.0              # seconds per sample-iteration (vectorised over 8)
    it = max(1, min(steps, 10))
    Bc = int(max(1, min(y_dev.shape[0], budget_s / max(per * it, 1e-9))))
    yc = y_dev[:Bc].double().cpu().numpy()
    t0 = time.perf_counter()
    o.combined_loop(yc, n_iters=it)
    dt = time.perf_counter() - t0
    return {"value": Bc * it / dt, "unit": "ADMM sample-iterations/s", "cores": 1, "kind": "port",
            "sample": f"{Bc} samples x {it} ADMM iterations of the same workload, float64, NumPy/SciPy oracle "
                      f"(vectorised over the batch, per-sample CG convergence), {dt:.1f} s on 1 of {os.cpu_count()} host cores"}


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(path))
        return d.get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def roofline_from_prof(prof, workload, path="stream", elems=0, cg=None):
    """Roofline object of the dominant kernel from the live HIP-event timings (tag 0).

    stream path: dominant kernel = the sparse-Laplacian SpMM inside the CG solves; algorithmic bytes are
                 accumulated per launch by the library (8 B/element per SpMM pass + CSR bytes).
    lds path   : dominant kernel = k_admm_lds (one launch = one whole ADMM iteration, CG loops inside LDS);
                 algorithmic bytes per launch follow SURVEY.md 8(d): U * [52 (Kx+1) + 44 (Kzu+1) + 52 (Kzd+1)
                 + 130] B with the measured mean CG counts.  These bytes never reach HBM on this path, so
                 `achieved` can exceed the HBM peak; `traffic` is the HBM traffic measured with rocprofv3."""
    p0 = prof[0]
    if p0["count"] == 0 or p0["ms"] <= 0:
        return None
    avg_ms = p0["ms"] / p0["count"]
    if path == "lds":
        per_elem = 52 * (cg["CG_iter_x"] + 1) + 44 * (cg["CG_iter_zu"] + 1) + 52 * (cg["CG_iter_zd"] + 1) + 130
        bytes_per = per_elem * elems
        kernel = "k_admm_lds<TPG> (LDS-resident fused ADMM iteration: 3 CG solves + prox + duals + history per launch)"
    else:
        bytes_per = p0["bytes"] / p0["count"]
        kernel = "k_tile / k_rows <SpMM in CG> (sparse mixed-graph Laplacian, batch-innermost; LDS-tiled on cluster-ordered graphs)"
    ach = bytes_per / (avg_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": ach / HBM_PEAK_GBS, "traffic": load_traffic(workload), "launches": p0["count"],
           "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": bytes_per}
    if path == "lds":
        out["note"] = ("algorithmic bytes (SURVEY 8d) are served from LDS/registers on this path; see `traffic` for the "
                       "HBM bytes actually moved and `roofline_cfg3` for the HBM-streaming SpMM kernel")
    else:
        out["cg_update_kernel_GBs"] = (prof[1]["bytes"] / max(prof[1]["ms"], 1e-9) / 1e6) if prof[1]["count"] else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg3-leg", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before the first HIP call (dmabuf IPC for RCCL)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # one process per GPU; MGADMM_DIST_BACKEND=gloo (+ several ranks sharing a GPU) exists only to rehearse the
    # multi-rank control flow on a one-GPU box
    backend = os.environ.get("MGADMM_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    cdev = device if backend == "nccl" else torch.device("cpu")      # where collectives take their tensors

    n, B, cl, dl, info, desc = build_problem(args.workload)
    if args.batch:
        B = args.batch
    blk = make_solver(n, cl, dl, info, device)
    y = synth_y(n, B, 12, seed=1, offset=rank * B, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(iters, prof):
        blk.max_ADMM_iter = iters
        blk._reset_history()
        if prof:
            blk.prof_begin()
        x = blk.combined_loop(y, print_info=False)
        gathered = None
        if world > 1:                       # the only exchange of the path: final gather of the x shards (RCCL/xGMI)
            xs = x.to(cdev)
            gathered = [torch.empty_like(xs) for _ in range(world)] if rank == 0 else None
            dist.gather(xs, gathered, dst=0)
        pr = blk.prof_end() if prof else None
        return x, pr

    # warm-up (untimed): creates the solver workspace and the HIP event pool, pages kernels in
    run(max(1, args.warmup), False)
    if not args.no_prof:
        run(1, True)                        # the event pool of mgadmm_prof_begin is created on first use
    barrier()
    t0 = time.perf_counter()
    x, prof = run(args.steps, not args.no_prof)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=cdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    cg_counts = {k2: float(torch.stack([v.float().mean() if torch.is_tensor(v) else torch.tensor(float(v))
                                        for v in getattr(blk, k2)]).mean()) for k2 in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd")}
    from mgadmm import _lib
    path = "lds" if _lib.lib.mgadmm_solver_path(blk._solvers[(1, torch.float32)][0], B) == _lib.PATH_LDS else "stream"
    finite = bool(torch.isfinite(x).all().item())

    out = None
    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "ADMM iters/sec x batch (sample-iterations/s)", "value": value,
            "unit": "ADMM sample-iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "batch_per_gpu": B, "global_batch": world * B,
                       "ablation": "None", "graph": "kNN-directed", "cg_tol": 1e-8, "max_cg_iter": 100,
                       "parallelism": f"batch-sharded x{world}, no collective on the convergence path, final RCCL gather",
                       "mean_cg_iters": cg_counts, "solver_path": path, "all_finite": finite,
                       "workspace_GB": blk.workspace_bytes() / 1e9, "per_kernel_events_in_timed_region": not args.no_prof},
        }
        if prof:
            out["roofline"] = roofline_from_prof(prof, args.workload, path, elems=B * 24 * n, cg=cg_counts)
        else:
            out["roofline"] = None

    # ---- CG-SpMV roofline leg on the 10k-node graph (BASELINE config 3), rank 0 of a 1-GPU run only
    if rank == 0 and world == 1 and args.workload == "cfg2" and not args.no_cfg3_leg:
        blk.close()
        del x
        torch.cuda.empty_cache()
        n3, B3, cl3, dl3, info3, desc3 = build_problem("cfg3")
        b3 = make_solver(n3, cl3, dl3, info3, device)
        y3 = synth_y(n3, B3, 12, seed=1, offset=0, device=device)
        b3.max_ADMM_iter = 1
        b3.combined_loop(y3, print_info=False)
        torch.cuda.synchronize()
        b3.max_ADMM_iter = 2
        b3._reset_history()
        b3.prof_begin()
        t0 = time.perf_counter()
        b3.combined_loop(y3, print_info=False)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        r3 = roofline_from_prof(b3.prof_end(), "cfg3")
        if r3:
            r3["config"] = f"cfg3: {desc3}, B={B3}, fp32, 2 ADMM iterations"
            r3["sample_iterations_per_s"] = B3 * 2 / dt3
        out["roofline_cfg3"] = r3
        b3.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            nb, Bb, clb, dlb, infob, _ = build_problem(args.workload)
            bb = make_solver(nb, clb, dlb, infob, device)
            yb = synth_y(nb, min(512, B), 12, seed=1, offset=0, device=device)
            out["cpu_baseline"] = cpu_baseline(nb, clb, infob, bb, yb, args.cpu_budget, args.steps)
            bb.close()
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line to it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
