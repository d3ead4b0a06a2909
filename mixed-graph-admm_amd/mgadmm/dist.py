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


def sharded_solve(solve_fn, y, mask=None, *, gather=True, group=None):
    """Run ``solve_fn(y_shard, mask_shard) -> x_shard`` on this rank's batch block and gather the shards.

    ``solve_fn`` is normally ``ADMM_algorithm.combined_loop`` bound to a solver on this rank's GPU.
    With ``gather=True`` every rank returns the full (B,T,N,C) tensor (all_gather of equal-size padded
    blocks); with ``gather=False`` each rank returns only its block.  Works without an initialised
    process group (world_size 1).
    """
    if dist.is_available() and dist.is_initialized():
        ws, rk = dist.get_world_size(group), dist.get_rank(group)
    else:
        ws, rk = 1, 0
    B = y.shape[0]
    lo, hi = shard_bounds(B, ws, rk)
    ys = y[lo:hi]
    ms = mask[lo:hi] if mask is not None else None
    xs = solve_fn(ys, ms) if hi > lo else None
    if not gather or ws == 1:
        return xs
    # equal-size blocks for all_gather: pad to the largest shard
    width = (B + ws - 1) // ws
    shape = None
    if xs is not None:
        shape = torch.tensor(list(xs.shape[1:]), dtype=torch.int64)
    # every rank has at least one sample unless B < world_size; handle the empty case by broadcasting the shape
    shp = [torch.zeros(3, dtype=torch.int64) for _ in range(ws)]
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = (shape if shape is not None else torch.zeros(3, dtype=torch.int64)).to(dev)
    shp = [s.to(dev) for s in shp]
    dist.all_gather(shp, mine, group=group)
    tail = next(tuple(int(v) for v in s.tolist()) for s in shp if int(s.sum()) > 0)
    dtype = xs.dtype if xs is not None else y.dtype
    buf = torch.zeros((width,) + tail, dtype=dtype, device=dev)
    if xs is not None:
        buf[: hi - lo] = xs.to(dev)
    parts = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r in range(ws):
        l2, h2 = shard_bounds(B, ws, r)
        out.append(parts[r][: h2 - l2])
    return torch.cat(out, 0).to(y.device if backend != "nccl" else dev)
