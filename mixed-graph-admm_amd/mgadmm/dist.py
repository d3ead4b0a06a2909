"""Multi-GPU execution: independent signals / time windows shard across the GPUs of one node.

Every sample is an independent optimisation (all CG reductions are per sample, reference
ADMM.py:347-356), so the batch is cut into contiguous blocks, one per rank (one process per GPU,
``torch.distributed`` with backend "nccl" = RCCL over xGMI, or "gloo" on CPU for tests), the graph
tables are replicated, and there is NO collective on the convergence path.  The only exchanges are at the
end: one gather of the x shards to the destination rank and one gather of the (tiny) per-shard residual
history, from which the reference-style whole-batch norms are re-formed (``gather_history``).

What sharding changes and what it does not
  * iterates: nothing, for a fixed iteration count (``check_stop=False`` or the same ``max_ADMM_iter`` reached
    everywhere) -- sample b's x is bitwise the x of the unsharded run;
  * early stop: the reference's stop test (ADMM.py:645-646) uses whole-BATCH norms.  Without a collective per
    iteration each shard can only test its own norms, so with ``check_stop=True`` a shard stops when ITS samples have
    converged (never later than the whole batch would).  Use a fixed iteration count when bit-identical results
    across different GPU counts matter; ``gather_history`` reports the iteration count of every shard.
    ``sharded_solve_global_stop`` keeps the reference's semantics instead: one iteration at a time on every rank and an
    all-reduce of six squared norms after each -- the stop test sees the whole-batch norms, the result is the one of the
    unsharded run, at the price of one tiny collective and one solver call per iteration.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world_size, rank):
    """Contiguous batch block [lo, hi) of `rank`; the first B % world_size ranks get one extra sample."""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _coll_device(group):
    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def _exchange_blocks(xs, n_local, width, tail, dtype, dev, ws, rk, group, mode, dst, async_op):
    """One equal-size padded block per rank -> every rank ('all') or the destination rank ('root').
    Returns (parts or None, work)."""
    buf = torch.zeros((width,) + tail, dtype=dtype, device=dev)
    if xs is not None and n_local > 0:
        buf[:n_local] = xs.to(dev)
    if mode == "all":
        parts = [torch.empty_like(buf) for _ in range(ws)]
        work = dist.all_gather(parts, buf, group=group, async_op=async_op)
        return parts, work
    parts = [torch.empty_like(buf) for _ in range(ws)] if rk == dst else None
    gdst = dist.get_global_rank(group, dst) if group is not None else dst
    work = dist.gather(buf, parts, dst=gdst, group=group, async_op=async_op)
    return parts, work


def sharded_solve(solve_fn, y, mask=None, *, gather="root", group=None, chunks=1, dst=0):
    """Run ``solve_fn(y_shard, mask_shard) -> x_shard`` on this rank's batch block and collect the shards.

    ``solve_fn`` is normally ``ADMM_algorithm.combined_loop`` bound to a solver on this rank's GPU.
      gather="root"            one gather to rank ``dst`` (the exchange BASELINE.json's north star names): rank ``dst``
                               returns the full (B,T,N,C) tensor, every other rank returns None (the default since
                               round 2; round 1 returned the full tensor everywhere);
      gather="all" (or True)   all_gather: every rank returns the full tensor (costs world_size x the memory) -- what
                               ``gather=True`` has always meant;
      gather=False             no exchange, each rank returns its own block.
    Works without an initialised process group (world_size 1).

    ``chunks > 1`` cuts the rank's block into that many sub-blocks that are solved one after the other; the
    exchange of sub-block c is issued asynchronously (RCCL runs it on its own stream) and overlaps the solve
    of sub-block c+1, so only the last sub-block's exchange is exposed.  Samples are independent, so for a fixed
    iteration count the result does not depend on ``chunks`` or on the number of ranks (see the module docstring
    for early stopping).
    """
    if gather is True:
        gather = "all"
    if gather not in ("root", "all", False):
        raise ValueError(f"gather must be 'root', 'all' or False, got {gather!r}")
    ws, rk = _world(group)
    B = y.shape[0]
    lo, hi = shard_bounds(B, ws, rk)
    chunks = max(1, int(chunks))
    if not gather or ws == 1:
        if hi <= lo:
            return None
        outs = []
        for c in range(chunks):
            a, b = shard_bounds(hi - lo, chunks, c)
            if b > a:
                outs.append(solve_fn(y[lo + a:lo + b], mask[lo + a:lo + b] if mask is not None else None))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)
    dev = _coll_device(group)
    width = (B + ws - 1) // ws                       # largest shard
    cwidth = (width + chunks - 1) // chunks          # largest sub-block of any rank
    pending = []                                     # (parts, work) per sub-block
    tail, dtype = None, None
    for c in range(chunks):
        a, b = shard_bounds(hi - lo, chunks, c)
        xs = None
        if b > a:
            xs = solve_fn(y[lo + a:lo + b], mask[lo + a:lo + b] if mask is not None else None)
        if tail is None:
            # trailing shape / dtype: every rank has at least one sample unless B < world_size
            mine = torch.zeros(4, dtype=torch.int64)
            if xs is not None:
                mine[:3] = torch.tensor(list(xs.shape[1:]), dtype=torch.int64)
                mine[3] = 1 if xs.dtype == torch.float64 else 0
            shp = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(ws)]
            dist.all_gather(shp, mine.to(dev), group=group)
            first = next(s for s in shp if int(s[:3].sum()) > 0)
            tail = tuple(int(v) for v in first[:3].tolist())
            dtype = xs.dtype if xs is not None else (torch.float64 if int(first[3]) else torch.float32)
        pending.append(_exchange_blocks(xs, b - a, cwidth, tail, dtype, dev, ws, rk, group, gather, dst,
                                        async_op=chunks > 1 and c < chunks - 1))
    for c in range(chunks):
        parts, work = pending[c]
        if work is not None:
            work.wait()
    if gather == "root" and rk != dst:
        return None
    out = []
    for r in range(ws):
        l2, h2 = shard_bounds(B, ws, r)
        for c in range(chunks):
            a, b = shard_bounds(h2 - l2, chunks, c)
            if b > a:
                out.append(pending[c][0][r][: b - a])
    return torch.cat(out, 0).to(y.device if dev.type == "cpu" else dev)


def sharded_solve_global_stop(blk, y, mask=None, *, gather="root", group=None, dst=0):
    """The reference's loop WITH its stop test on a sharded batch (ADMM.py:546-646: the test uses whole-batch norms).

    ``blk`` is this rank's ``ADMM_algorithm``.  Every rank runs its block one ADMM iteration at a time (``solve`` resumed from
    the state of the previous call: k calls of one iteration are bitwise one call of k iterations), the squared primal / dual
    residual norms of the iteration are all-reduced (sum: a Frobenius norm over the batch is the root of the sum of the
    shards' squares) and every rank applies ``max(primal) < ADMM_tol and max(dual) < ADMM_tol`` (ADMM.py:645) to the same
    whole-batch values: all ranks stop after the same iteration, which is the iteration the unsharded run stops after.
    A NaN / Inf on any rank (the reference's asserts, ADMM.py:534-606) is raised on every rank after the same iteration.
    ``blk``'s history lists hold the per-shard values of the iterations that ran (``gather_history`` re-forms the whole-batch
    ones).  Returns ``(x, n_iterations)``; ``x`` as ``sharded_solve`` returns it for the same ``gather``.
    """
    ws, rk = _world(group)
    B = y.shape[0]
    lo, hi = shard_bounds(B, ws, rk)
    ys = y[lo:hi]
    ms = mask[lo:hi] if mask is not None else None
    ncomp = 1 + int(blk.ablation in ("None", "DGLR")) + int(blk.ablation != "DGLR")
    n_max, save_stop = int(blk.max_ADMM_iter), blk.check_stop
    tol = float(blk.ADMM_tol)
    dev = _coll_device(group) if ws > 1 else torch.device("cpu")
    x, state, n_it, failure = None, None, 0, None
    blk.max_ADMM_iter, blk.check_stop = 1, False
    try:
        for it in range(n_max):
            loc = torch.zeros(2 * ncomp + 1, dtype=torch.float64)
            if hi > lo and failure is None:
                try:
                    x = blk.solve(ys, mask=ms, return_state=True, warm_start=state)[0]
                    state = blk.state
                    loc[:ncomp] = torch.tensor([float(v) ** 2 for v in blk.p_res_list[-1]], dtype=torch.float64)
                    loc[ncomp:2 * ncomp] = torch.tensor([float(v) ** 2 for v in blk.d_res_list[-1]], dtype=torch.float64)
                except AssertionError as e:          # NaN / Inf in this shard: tell the others, then raise everywhere
                    failure = e
                    loc[2 * ncomp] = 1.0
            if ws > 1:
                t = loc.to(dev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                loc = t.cpu()
            n_it = it + 1
            if float(loc[2 * ncomp]) > 0:
                raise failure if failure is not None else AssertionError(
                    "NaN/Inf value in the ADMM iterates of another shard (reference asserts, ADMM.py:534-606)")
            pri, dual = torch.sqrt(loc[:ncomp]), torch.sqrt(loc[ncomp:2 * ncomp])
            if float(pri.max()) < tol and float(dual.max()) < tol:
                break
    finally:
        blk.max_ADMM_iter, blk.check_stop = n_max, save_stop
    if not gather or ws == 1:
        return x, n_it
    return sharded_solve(lambda _y, _m: x, y, mask, gather=gather, group=group, dst=dst), n_it


# ---------------------------------------------------------------------------------------------- residual history
_NORM_KEYS = ("p_res_list", "d_res_list", "x_shift_list", "recover_list")        # Frobenius norms over the batch
_MEAN_KEYS = ("GLR_list", "DGTV_list", "DGLR_list")                             # means over the samples


def _as_matrix(v):
    if len(v) == 0:
        return torch.zeros((0, 0), dtype=torch.float64)
    t = torch.as_tensor([[float(e) for e in row] if isinstance(row, (list, tuple)) else [float(row)] for row in v],
                        dtype=torch.float64)
    return t


def _last_rows(m, rows):
    """The last `rows` rows of a history matrix (a list that survives ``init_iterations`` -- recover_list -- is longer than
    the others: the rows of the solve just finished are its last ones)."""
    return m[m.shape[0] - rows:] if m.shape[0] >= rows else m


def gather_history(history, n_local, *, group=None, dst=0, chunks=1):
    """Whole-batch residual history of a sharded solve, re-formed on rank ``dst`` from the per-shard histories
    (SURVEY.md 8e): norms combine as sqrt(sum_shards local^2) (ADMM.py:612-636 are Frobenius norms over the batch
    tensor), regularisers (means over samples, ADMM.py:230-246) as sample-weighted means.  One gather of a few hundred
    floats, after the loop -- nothing on the convergence path.

    ``history`` is ``ADMM_algorithm.history()`` (or the instance itself) of this rank's solve of ``n_local`` samples.
    Ranks may have executed different iteration counts (early stop is per shard): the combined lists cover the
    iterations EVERY non-empty shard executed; ``iters_per_shard`` holds each shard's own count.  ``delta_x_per_step`` is
    the norm of a batch MEAN (ADMM.py:614) and cannot be combined from shard norms: it is returned per shard.
    Only the rows of the LAST solve of every list enter (a list may hold older rows: ``recover_list`` survives
    ``init_iterations`` like in the reference); every list travels with its own row count.
    ``chunks``: the value given to ``sharded_solve(chunks=)`` -- the rank's history then holds one block of rows per
    sub-block (all with the same iteration count: use a fixed ``max_ADMM_iter``); the blocks are combined like shards
    before the exchange, ``delta_x_per_step_per_shard`` keeps one block per sub-block.
    Returns a dict on rank ``dst`` and None elsewhere.
    """
    get = (lambda k: history[k]) if isinstance(history, dict) else (lambda k: getattr(history, k))
    ws, rk = _world(group)
    chunks = max(1, int(chunks))
    mats = {k: _as_matrix(get(k)) for k in _NORM_KEYS + _MEAN_KEYS}
    dxps = torch.stack([torch.as_tensor(v, dtype=torch.float64) for v in get("delta_x_per_step")]) \
        if len(get("delta_x_per_step")) else torch.zeros((0, 0), dtype=torch.float64)
    rows = mats["p_res_list"].shape[0]
    if chunks > 1 and n_local > 0:
        nblk = sum(1 for c in range(chunks) if shard_bounds(n_local, chunks, c)[1] > shard_bounds(n_local, chunks, c)[0])
        if nblk == 0 or rows % nblk:
            raise ValueError(f"gather_history(chunks={chunks}): {rows} history rows are not {nblk} blocks of equal length "
                             "(sub-blocks must run the same iteration count: check_stop=False)")
        iters = rows // nblk
        sizes_c = [shard_bounds(n_local, chunks, c)[1] - shard_bounds(n_local, chunks, c)[0] for c in range(chunks)]
        sizes_c = [v for v in sizes_c if v > 0]
        for k in mats:
            m = _last_rows(mats[k], rows)
            blocks = [m[j * iters:(j + 1) * iters] for j in range(nblk)]
            if k in _NORM_KEYS:
                mats[k] = torch.sqrt(sum(b ** 2 for b in blocks))
            else:
                mats[k] = sum(b * n for b, n in zip(blocks, sizes_c)) / float(n_local)
    else:
        iters = rows
        mats = {k: _last_rows(mats[k], iters) for k in mats}
    dxps = _last_rows(dxps, rows)
    if ws == 1:
        out = {k: mats[k] for k in mats}
        out.update(iters_per_shard=[iters], samples_per_shard=[n_local], delta_x_per_step_per_shard=[dxps])
        return out
    dev = _coll_device(group)
    # fixed-size packet per rank: [n_local, iters, (rows, cols) per key ..., payload padded to the longest]
    keys = list(mats) + ["delta_x_per_step"]
    vals = [mats[k] for k in mats] + [dxps]
    shape = []
    for v in vals:
        shape += [v.shape[0], v.shape[1] if v.numel() else 0]
    head = torch.tensor([n_local, iters] + shape, dtype=torch.float64)
    body = torch.cat([v.reshape(-1) for v in vals])
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, torch.tensor([head.numel() + body.numel()], dtype=torch.int64, device=dev), group=group)
    width = int(max(int(s) for s in sizes))
    pkt = torch.zeros(width, dtype=torch.float64, device=dev)
    pkt[: head.numel() + body.numel()] = torch.cat([head, body]).to(dev)
    parts = [torch.empty_like(pkt) for _ in range(ws)] if rk == dst else None
    gdst = dist.get_global_rank(group, dst) if group is not None else dst
    dist.gather(pkt, parts, dst=gdst, group=group)
    if rk != dst:
        return None
    shards = []
    for p in parts:
        p = p.cpu()
        nl, it = int(p[0]), int(p[1])
        off = 2 + 2 * len(keys)
        d = {}
        for j, k in enumerate(keys):
            r, c = int(p[2 + 2 * j]), int(p[3 + 2 * j])
            d[k] = p[off: off + r * c].reshape(r, c) if c else torch.zeros((r, 0), dtype=torch.float64)
            off += r * c
        shards.append((nl, it, d))
    live = [s for s in shards if s[0] > 0]
    n_it = min(s[1] for s in live) if live else 0
    total = sum(s[0] for s in live)
    out = {}
    for k in _NORM_KEYS:
        out[k] = torch.sqrt(sum(s[2][k][:n_it] ** 2 for s in live)) if live else torch.zeros((0, 0))
    for k in _MEAN_KEYS:
        out[k] = sum(s[2][k][:n_it] * s[0] for s in live) / max(total, 1) if live else torch.zeros((0, 0))
    out["iters_per_shard"] = [s[1] for s in shards]
    out["samples_per_shard"] = [s[0] for s in shards]
    out["delta_x_per_step_per_shard"] = [s[2]["delta_x_per_step"] for s in shards]
    return out
