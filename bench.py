#!/usr/bin/env python3
"""Benchmark of the ADMM hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one ADMM iteration (3 batched CG solves + prox + dual updates + residual history,
reference ADMM.py:546-646) over the whole batch resident in HBM.  The timed region is ONE solve of
exactly K iterations (no early stop), including its small set-up (layout pack, initial guess) and the
final unpack / gather, bracketed by barrier + torch.cuda.synchronize(); the time is the MAX over ranks.
With N > 1 the only exchange of the path -- the gather of the x shards to rank 0 -- is inside the timed region; with
--gather-chunks c a rank solves its block as c sub-blocks and the asynchronous gather of one overlaps the solve of the next.

Workloads (BASELINE.json configs; synthetic data, seeds fixed):
  cfg2 (default) PEMS04-shaped graph: N=307, 340 undirected edges, k=4, B=4096 windows per GPU, fp32
  cfg3           N=10 000 Euclidean kNN graph, B=512 per GPU, fp32     (CG-SpMV HBM roofline run)
  cfg4           N=100 000 mixed graph, B=256 per GPU (weak scaling 1..8 GPUs), fp32
Metric: ADMM sample-iterations per second = (samples over all ranks) * K / seconds.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the timed workload, measured live with
HIP events recorded around every launch on the launch stream during the timed solve:
  * cfg2 (LDS-resident path): k_admm_lds never streams its vectors from HBM, so it is priced against the LDS:
    `bound` = "lds", `achieved` = LDS bytes the kernel's gathers / stores move per launch (exact count from the table
    widths, the time-group width and the CG iteration counts of the timed solve; a launch runs several ADMM iterations:
    `iterations_per_launch`) / average launch duration, `peak` = 150 TB/s (MI355X_MICROARCH.md: aggregate ds_read_b128
    rate); `traffic` = HBM bytes per launch from the rocprofv3 PMC passes (profiles/traffic.json holds them per ADMM
    iteration), `hbm_achieved` = traffic / duration.
  * cfg3 / cfg4 (streaming path): the sparse-Laplacian SpMM inside the CG solves against the HBM roofline
    (`bound` = "hbm", 8 TB/s), algorithmic bytes per launch as accumulated by the library.
The default (cfg2) run also measures the cfg3 SpMM roofline -- the configuration the >= 60 % target is quoted on -- over
6 ADMM iterations (> 200 live launches); its figures are flattened into the same `roofline` object as `cfg3_spmm_*`, and a
2-iteration leg on the 100k-node graph of cfg4 (B = 256, the per-GPU slice) is reported as `roofline_cfg4`.
`device_state` (and `roofline_cfg3/4.device_state`; only with --device-state): engine / memory clock and board power from sysfs.
`cpu_baseline` is the CPU oracle (NumPy/SciPy restatement of the reference) on a bounded sample of the same windows:
one worker process per host core (at most 16, the GPU box's CPU share), each solving its windows one at a time
(B = 1: the reference's semantics), plus the vectorised-batch rate of one core and the full cfg1 run.
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

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
LDS_PEAK_GBS = 150000.0  # aggregate ds_read_b64/b128 rate with every CU streaming (MI355X_MICROARCH.md, LDS section)


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


class DeviceSampler:
    """Engine clock / memory clock / board power of the GPU around a timed region, read from the amdgpu hwmon files of sysfs (no
    HIP call, no rocm-smi process): by a side thread every 50 ms during the streaming legs, and ONCE right after the timed
    region of the headline workload.  OPT-IN (--device-state): runs that looked the card up in sysfs before the timed region
    measured k_admm_lds 4-5 % slower (1 766 vs 1 845 us per iteration, alternating runs on three boxes) whether or not
    anything was read during the region -- the cause was not found (waking the other cards of the host through their hwmon
    directories is the suspect), so the default bench line touches no sysfs file.  With the flag the JSON line carries
    `device_state`, so that a box that runs slow can be told from a code change (round 2: one box ran the same launch 17 %
    slower).  The card is found by the PCI address of the HIP device; if that fails every amdgpu card the container shows is
    sampled and the BUSIEST one (highest mean power) is reported.  Silently absent when the files are not there."""

    KEYS = (("sclk_mhz", "freq1_input"), ("mclk_mhz", "freq2_input"), ("power_w", "power1_average"), ("power_w", "power1_input"))

    @staticmethod
    def pci_address(device_index):
        """'dddd:bb:dd.f' of a HIP device (to find its sysfs card), or None."""
        try:
            p = torch.cuda.get_device_properties(device_index)
            return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            return None

    def __init__(self, pci=None, period=0.05):
        import glob
        self.period = period     # seconds between reads of the side thread; None: no thread, read_once() only
        self.cards = []          # one {key: file} per card
        self.pci = pci
        self.matched = False
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            if pci is not None and os.path.basename(os.path.realpath(c)).lower() != pci.lower():
                continue
            try:
                if open(os.path.join(c, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            files = {}
            for hw in glob.glob(os.path.join(c, "hwmon", "hwmon*")):
                for key, name in self.KEYS:
                    f = os.path.join(hw, name)
                    if key not in files and os.path.exists(f):
                        files[key] = f
            if files:
                self.cards.append(files)
        if pci is not None and self.cards:
            self.matched = True
        elif pci is not None:            # address not found among the sysfs cards: sample all of them, report the busiest
            self.__init__(None, period)
            self.pci = pci
            return
        self.samples = [{k: [] for k in f} for f in self.cards]
        self._stop = False
        self._thread = None

    def _read(self):
        for files, smp in zip(self.cards, self.samples):
            for k, f in files.items():
                try:
                    smp[k].append(float(open(f).read().strip()) / 1e6)          # Hz -> MHz, uW -> W
                except (OSError, ValueError):
                    pass

    def read_once(self):
        self._read()

    def __enter__(self):
        import threading
        if self.cards and self.period:
            def loop():
                while not self._stop:
                    self._read()
                    time.sleep(self.period)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def summary(self):
        best, best_p = None, -1.0
        for smp in self.samples:
            pw = smp.get("power_w") or smp.get("sclk_mhz") or []
            m = sum(pw) / len(pw) if pw else 0.0
            if m > best_p:
                best, best_p = smp, m
        if not best:
            return None
        out = {"cards_sampled": len(self.samples), "card": self.pci if self.matched else "busiest of the sampled cards"}
        for k, v in best.items():
            if v:
                out[k] = {"min": round(min(v), 1), "mean": round(sum(v) / len(v), 1), "max": round(max(v), 1), "samples": len(v)}
        return out


def make_solver(n, cl, dl, info, device):
    import mgadmm
    blk = mgadmm.ADMM_algorithm({"n_nodes": n}, info, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl, dl),
                                device=device, compute_dtype=torch.float32, record_cg_coeffs=False)
    blk.check_stop = False          # fixed iteration count: exactly K steps
    return blk


def _oracle_inputs(cl, blk, info, y_dev, path):
    np.savez(path, cl=cl.numpy(), u_ew=blk.u_ew[0].numpy(), d_ew=blk.d_ew[0].numpy(), y=y_dev.double().cpu().numpy(),
             t_in=12, T=24, **{k: float(v) for k, v in info.items()})


def _run_workers(npz, bounds, mode, iters):
    """Start one oracle worker per (lo, hi) slice, release them together, return (wall seconds, samples)."""
    import subprocess
    worker = os.path.join(ROOT, "oracle", "cpu_bench_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, npz, str(lo), str(hi), mode, str(iters)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True) for lo, hi in bounds]
    for p in procs:
        assert p.stdout.readline().strip() == "READY"
    t0 = time.perf_counter()
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()
    outs = [json.loads(p.stdout.readline()) for p in procs]
    dt = time.perf_counter() - t0
    for p in procs:
        p.wait()
    return dt, sum(o["samples"] for o in outs)


def cpu_baseline(workload, n, cl, info, blk, y_dev, budget_s, steps):
    """Oracle (CPU restatement of the reference algorithm) on a bounded sample of the same workload, per
    BASELINE.md section 4: every host core the box gives one GPU job (<= 16 worker processes, one core each), the
    per-sample loop (B = 1, the reference's own semantics) AND the vectorised batch; cfg1 timed in full."""
    import tempfile
    from oracle import admm_oracle as orc
    ncpu = os.cpu_count() or 1
    workers = max(1, min(16, ncpu))
    tmp = tempfile.mkdtemp(prefix="mgadmm_cpu_")
    # calibrate one core (vectorised, 8 windows x 2 iterations), then size the sample to ~budget_s/2 per mode
    o = orc.OracleADMM(cl.numpy(), blk.u_ew[0].numpy(), blk.d_ew[0].numpy(), info, mode="knn", t_in=12, T=24)
    ncal = min(8, y_dev.shape[0])
    yc0 = y_dev[:ncal].double().cpu().numpy()
    t0 = time.perf_counter()
    o.combined_loop(yc0, n_iters=2)
    per = (time.perf_counter() - t0) / (2.0 * ncal)                 # seconds per sample-iteration, one core, vectorised
    it = max(1, min(steps, 50))          # the same iteration count as the GPU line (later iterations are cheaper: fewer CG iterations)
    nb = int(max(1, min(y_dev.shape[0], 64, (budget_s / 3) / max(per * it, 1e-9))))      # vectorised batch on one core
    nl = int(max(workers, min(y_dev.shape[0], workers * (budget_s / 2) / max(per * it, 1e-9))))
    npz = os.path.join(tmp, "in.npz")
    _oracle_inputs(cl, blk, info, y_dev[:max(nb, nl)], npz)
    from mgadmm.dist import shard_bounds
    dt_loop, ns = _run_workers(npz, [shard_bounds(nl, workers, w) for w in range(workers) if shard_bounds(nl, workers, w)[1] > shard_bounds(nl, workers, w)[0]], "loop", it)
    nw = min(workers, nl)
    dt_vec, nv = _run_workers(npz, [(0, nb)], "batch", it)
    out = {"value": ns * it / dt_loop, "unit": "ADMM sample-iterations/s", "cores": nw, "kind": "port",
           "sample": f"{ns} windows of the same workload x {it} ADMM iterations, float64 NumPy/SciPy oracle, one window at a "
                     f"time (B=1, reference semantics) on {nw} worker processes = {nw} of {ncpu} host cores, {dt_loop:.1f} s",
           "host_cpu_count": ncpu,
           "vectorised_1core_value": nv * it / dt_vec,
           "vectorised_1core_sample": f"{nv} windows x {it} iterations as one batch on 1 core, {dt_vec:.1f} s"}
    if workload != "cfg1":
        try:
            n1, _, cl1, dl1, info1, _ = build_problem("cfg1")
            import mgadmm
            b1 = mgadmm.ADMM_algorithm({"n_nodes": n1}, info1, use_kNN=True, k=4, u_sigma=50, d_sigma=50, tables=(cl1, dl1))
            y1 = 300 * torch.rand(1, 12, n1, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
            npz1 = os.path.join(tmp, "cfg1.npz")
            _oracle_inputs(cl1, b1, info1, y1, npz1)
            b1.close()
            dt1, _ = _run_workers(npz1, [(0, 1)], "loop", 50)
            out["cfg1_full_run"] = f"N=170, B=1, float64, 50 ADMM iterations on 1 core: {dt1:.2f} s = {50 / dt1:.1f} ADMM it/s"
            out["cfg1_full_run_it_per_s"] = 50 / dt1
        except Exception as e:  # noqa: BLE001
            out["cfg1_full_run"] = f"failed: {e!r}"
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(path))
        return d.get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def lds_bytes_per_launch(blk, B, cg, T=24):
    """LDS bytes ONE ADMM iteration of B samples moves in k_admm_lds, counted from what the kernel's element-owning threads
    issue (csrc/lds_kernels.h, round-3 form).  run = 4 TPG bytes = the aligned window of a gathered row.
      cLdr application inside a CG solve: W_d entries x run + 16 (v of the next time group) ; (lead + 2 tail_pairs) W_d^T
        positions x run (the padded tail positions are read like real ones) + 16 tail_pairs (the tail table entries) + 16 ;
        stores of q and p: 2 run.  Uniform instances keep the diagonal entries and the table rows in registers: W_d has
        ND - 1 gathered entries, no entry reads; the other instances read 8 B per table entry.
      Lu application: W_u entries x run (+ 8 B per entry: generic instances) + the store of p.
      Per iteration besides the solves: RHS_x (one W_d^T application on an image + its own-row read + the image store),
        phi prox (W_d) and GLR (W_u).
    Operator applications: x / zd solves K + 1 each, zu solve K + 1.  K = measured mean CG iterations of the timed solve."""
    from mgadmm import _lib
    h = blk._solvers[(1, torch.float32)][0]
    q = lambda w: _lib.query(h, w)
    tpg, nt = q(_lib.Q_LDS_TPG), q(_lib.Q_LDS_THREADS)
    uni, tp, lead = q(_lib.Q_LDS_UNIFORM), q(_lib.Q_LDS_TAIL_PAIRS), q(_lib.Q_LDS_LEAD)
    n = blk.n_nodes
    nu, nd = q(_lib.Q_NNZ_U) / n, q(_lib.Q_NNZ_D) / n          # entries per row
    run = 4 * tpg
    edge = 16 if tpg % 4 == 0 else 4
    ent = 0 if uni else 8                                      # table entry reads per gathered entry
    wd = ((nd - 1) if uni else nd) * (run + ent) + edge
    wdt = (lead + 2 * tp) * run + (0 if uni else lead * 8) + 2 * tp * 8 + edge
    wu = nu * (run + ent)
    cldr = wd + wdt + 2 * run
    lu = wu + run
    kx, kzu, kzd = cg["CG_iter_x"], cg["CG_iter_zu"], cg["CG_iter_zd"]
    per_thread = ((kx + 1) + (kzd + 1)) * cldr + (kzu + 1) * lu + (wdt + 2 * run) + (wd + 2 * run) + (wu + run)
    return B * nt * per_thread, tpg


def roofline_from_prof(prof, workload, path="stream", blk=None, B=0, cg=None, steps=0):
    """Roofline object of the dominant kernel from the live HIP-event timings (tag 0); see the module docstring."""
    p0 = prof[0]
    if p0["count"] == 0 or p0["ms"] <= 0:
        return None
    avg_ms = p0["ms"] / p0["count"]
    traffic = load_traffic(workload)
    if path == "lds":
        bytes_it, tpg = lds_bytes_per_launch(blk, B, cg)
        ipl = steps / p0["count"] if steps else 1.0       # ADMM iterations per launch (chunked schedule: a workgroup keeps its sample)
        bytes_per = bytes_it * ipl
        if traffic:
            traffic = traffic * ipl                       # profiles/traffic.json holds HBM bytes per ITERATION of the batch
        ach = bytes_per / (avg_ms * 1e-3) / 1e9
        out = {"bound": "lds", "kernel": f"k_admm_lds<TPG={tpg}> (LDS-resident fused ADMM iterations: 3 CG solves + prox + duals + "
                                         "history per iteration; one workgroup per sample, several iterations per launch)",
               "achieved": ach, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": ach / LDS_PEAK_GBS, "traffic": traffic,
               "launches": p0["count"], "avg_launch_us": avg_ms * 1e3, "iterations_per_launch": ipl,
               "avg_iteration_us": avg_ms * 1e3 / ipl, "lds_bytes_per_launch": bytes_per,
               "hbm_achieved": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
               "hbm_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
               "note": "vectors live in LDS/registers inside the CG solves: priced against the LDS pipe; `traffic`/`hbm_*` = HBM bytes "
                       "per launch from rocprofv3 PMC (state in/out only); the HBM-streaming SpMM roofline is in cfg3_spmm_*"}
        return out
    bytes_per = p0["bytes"] / p0["count"]
    ach = bytes_per / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "sparse mixed-graph Laplacian SpMM inside CG: k_cldr (fused Ldr^T Ldr with the CG vector update folded "
                                      "into its loads: x / zd solves) and k_tile (Lu: zu solve), batch-innermost, LDS-tiled",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "launches": p0["count"], "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": bytes_per,
            "cg_update_kernel_GBs": (prof[1]["bytes"] / max(prof[1]["ms"], 1e-9) / 1e6) if prof[1]["count"] else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="ADMM iterations of the timed solve (default 50: the iteration count of BASELINE config 1)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg3-leg", action="store_true", help="skip the streaming-path legs (cfg3 and cfg4)")
    ap.add_argument("--no-cfg4-leg", action="store_true", help="skip the 100k-node leg only")
    ap.add_argument("--device-state", action="store_true",
                    help="report engine / memory clock and board power from sysfs as `device_state` (off by default: a run that "
                         "merely looks the card up in sysfs before the timed region measured 4-5 %% slower on cfg2 -- 1 766 vs "
                         "1 845 us per iteration of k_admm_lds, alternating runs on three boxes; the reads themselves come after)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--gather-chunks", type=int, default=1,
                    help="N > 1: sub-blocks per rank; the final gather of one overlaps the solve of the next "
                         "(default 1: two half-batch solves cost 3.7 %% more than one at cfg2 -- 1.585 vs 1.646 M "
                         "sample-iterations/s, fixed per-solve work -- which is more than the overlap hides of a 120 MB/rank "
                         "gather; the streaming workloads would also drop to a narrower vector width)")
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

    # N > 1: the rank's block is solved as `chunks` sub-blocks one after the other (samples are independent: the same
    # iterates, mgadmm.dist.sharded_solve(chunks=) does the same); the gather of sub-block c is issued asynchronously
    # (RCCL runs it on its own stream) and overlaps the solve of sub-block c+1, so only the last sub-block's share of the
    # exchange is exposed.  N = 1 has no exchange and solves its block in one piece.  Default: one piece everywhere.
    from mgadmm.dist import shard_bounds
    chunks = max(1, min(args.gather_chunks, B)) if world > 1 else 1
    blocks = [bd for bd in (shard_bounds(B, chunks, c) for c in range(chunks)) if bd[1] > bd[0]]     # non-empty sub-blocks
    Bc = max(b - a for a, b in blocks)

    def run(iters, prof):
        blk.max_ADMM_iter = iters
        blk._reset_history()
        if prof:
            blk.prof_begin()
        if world == 1:
            return blk.combined_loop(y, print_info=False), None
        pending, x = [], None
        for c, (a, b) in enumerate(blocks):  # the only exchange of the path: final gather of the x shards (RCCL/xGMI)
            x = blk.combined_loop(y[a:b], print_info=False)
            xs = x.to(cdev)
            gathered = [torch.empty_like(xs) for _ in range(world)] if rank == 0 else None
            work = dist.gather(xs, gathered, dst=0, async_op=c + 1 < len(blocks))
            pending.append((xs, gathered, work))      # the buffers stay referenced until the exchange has completed
        for _, _, work in pending:
            if work is not None:
                work.wait()
        return x, None

    # warm-up (untimed): creates the solver workspace and the HIP event pool, pages kernels in
    run(max(1, args.warmup), False)
    run(args.steps, not args.no_prof)       # a solve of the timed length allocates whatever depends on the iteration count (iterate
    if not args.no_prof:                    # buffers of the chunked schedule); the event pool of mgadmm_prof_begin is created on first use
        blk.prof_end()
    sampler = DeviceSampler(DeviceSampler.pci_address(local), period=None) if rank == 0 and args.device_state else None
    barrier()
    t0 = time.perf_counter()
    x, prof = run(args.steps, not args.no_prof)
    barrier()
    dt = time.perf_counter() - t0
    if sampler is not None:
        sampler.read_once()          # clocks / averaged power right after the timed region (never during it: see DeviceSampler)
    # the HIP events were recorded around every launch INSIDE the timed region; their elapsed times are read after it
    t1 = time.perf_counter()
    prof = blk.prof_end() if not args.no_prof else None
    prof_read_ms = (time.perf_counter() - t1) * 1e3 if not args.no_prof else 0.0
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
                       "parallelism": f"batch-sharded x{world}, no collective on the convergence path, final RCCL gather"
                                      + (f" in {chunks} sub-blocks (all but the last overlap the next sub-block's solve)"
                                         if world > 1 else ""),
                       "mean_cg_iters": cg_counts, "solver_path": path, "all_finite": finite,
                       "workspace_GB": blk.workspace_bytes() / 1e9, "per_kernel_events_in_timed_region": not args.no_prof,
                       "event_readback_ms_after_timed_region": round(prof_read_ms, 3)},
        }
        out["config"]["mean_cg_iters_x_zu_zd"] = [round(cg_counts[k2], 2) for k2 in ("CG_iter_x", "CG_iter_zu", "CG_iter_zd")]
        out["device_state"] = sampler.summary() if sampler is not None else None
        out["roofline"] = roofline_from_prof(prof, args.workload, path, blk=blk, B=Bc, cg=cg_counts, steps=args.steps * len(blocks)) if prof else None

    # ---- CG-SpMV roofline legs on the streaming path, rank 0 of a 1-GPU run only: cfg3 (10k-node graph, BASELINE config 3:
    # the configuration the >= 60 % target is quoted on) and a short cfg4 leg (100k-node graph, the per-GPU slice of BASELINE
    # config 4) so that a driver-timed number exists for the largest graph
    def streaming_leg(name, iters):
        n3, B3, cl3, dl3, info3, desc3 = build_problem(name)
        b3 = make_solver(n3, cl3, dl3, info3, device)
        y3 = synth_y(n3, B3, 12, seed=1, offset=0, device=device)
        b3.max_ADMM_iter = 1
        b3.combined_loop(y3, print_info=False)
        torch.cuda.synchronize()
        b3.max_ADMM_iter = iters
        b3._reset_history()
        b3.prof_begin()
        import contextlib
        with (DeviceSampler(DeviceSampler.pci_address(local)) if args.device_state else contextlib.nullcontext()) as smp3:
            t0 = time.perf_counter()
            b3.combined_loop(y3, print_info=False)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
        r3 = roofline_from_prof(b3.prof_end(), name)
        if r3:
            r3["device_state"] = smp3.summary() if smp3 is not None else None
            r3["config"] = f"{name}: {desc3}, B={B3}, fp32, {iters} ADMM iterations"
            r3["sample_iterations_per_s"] = B3 * iters / dt3
        b3.close()
        del b3, y3
        torch.cuda.empty_cache()
        return r3

    if rank == 0 and world == 1 and args.workload == "cfg2" and not args.no_cfg3_leg:
        blk.close()
        del x
        torch.cuda.empty_cache()
        r3 = streaming_leg("cfg3", 6)             # 6 ADMM iterations: > 200 live SpMM-in-CG launches
        if r3 and out.get("roofline"):
            for k3 in ("bound", "achieved", "peak", "frac", "traffic", "launches", "avg_launch_us",
                       "algorithmic_bytes_per_launch", "cg_update_kernel_GBs", "sample_iterations_per_s"):
                out["roofline"]["cfg3_spmm_" + k3] = r3[k3]
        out["roofline_cfg3"] = r3
        if not args.no_cfg4_leg:
            try:
                out["roofline_cfg4"] = streaming_leg("cfg4", 2)
            except Exception as e:  # noqa: BLE001  (52 GB workspace: never lose the line to it)
                out["roofline_cfg4"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            nb, Bb, clb, dlb, infob, _ = build_problem(args.workload)
            bb = make_solver(nb, clb, dlb, infob, device)
            yb = synth_y(nb, min(1024, B), 12, seed=1, offset=0, device=device)
            out["cpu_baseline"] = cpu_baseline(args.workload, nb, clb, infob, bb, yb, args.cpu_budget, args.steps)
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
