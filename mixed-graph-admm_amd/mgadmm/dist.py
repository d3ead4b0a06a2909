"""Multi-GPU execution: independent signals / time windows shard across the GPUs of one node.

Every sample is an independent optimisation (all CG reductions are per sample, reference
ADMM.py:347-356), so the batch is cut into contiguous blocks, one per rank (one process per GPU,
``torch.distributed`` with backend "nccl" = RCCL over xGMI, or "gloo" on CPU for tests), the graph
tables are replicated, and there is NO collective on the convergence path.  The only exchange is the
final gather of the x shards (and of the per-shard residual history).
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world_size, rank):
    """Contiguous batch block [lo, hi) of `rank`; the first B % world_size ranks get one extra sample."""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _gather_blocks(xs, n_local, width, tail, dtype, dev, ws, group, async_op):
    """all_gather of one equal-size padded block per rank; returns (parts, work)."""
    buf = torch.zeros((width,) + tail, dtype=dtype, device=dev)
    if xs is not None and n_local > 0:
        buf[:n_local] = xs.to(dev)
    parts = [torch.empty_like(buf) for _ in range(ws)]
    work = dist.all_gather(parts, buf, group=group, async_op=async_op)
    return parts, work


def sharded_solve(solve_fn, y, mask=None, *, gather=True, group=None, chunks=1):
    """Run ``solve_fn(y_shard, mask_shard) -> x_shard`` on this rank's batch block and gather the shards.

    ``solve_fn`` is normally ``ADMM_algorithm.combined_loop`` bound to a solver on this rank's GPU.
    With ``gather=True`` every rank returns the full (B,T,N,C) tensor (all_gather of equal-size padded
    blocks); with ``gather=False`` each rank returns only its block.  Works without an initialised
    process group (world_size 1).

    ``chunks > 1`` cuts the rank's block into that many sub-blocks that are solved one after the other; the
    all_gather of sub-block c is issued asynchronously (RCCL runs it on its own stream) and overlaps the solve
    of sub-block c+1, so only the last sub-block's exchange is exposed.  Samples are independent, so the result
    does not depend on ``chunks``.
    """
    if dist.is_available() and dist.is_initialized():
        ws, rk = dist.get_world_size(group), dist.get_rank(group)
    else:
        ws, rk = 1, 0
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
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
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
        pending.append(_gather_blocks(xs, b - a, cwidth, tail, dtype, dev, ws, group, async_op=chunks > 1 and c < chunks - 1))
    out = []
    for r in range(ws):
        l2, h2 = shard_bounds(B, ws, r)
        for c in range(chunks):
            a, b = shard_bounds(h2 - l2, chunks, c)
            parts, work = pending[c]
            if work is not None:
                work.wait()
                pending[c] = (parts, None)
            if b > a:
                out.append(parts[r][: b - a])
    return torch.cat(out, 0).to(y.device if backend != "nccl" else dev)
