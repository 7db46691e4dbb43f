"""Batch sharding over the GPUs of one node.

Controller instances are independent: a tick needs no exchange between them, so the only communication of a
multi-GPU job is the one-time split of the seeded inputs (rank 0 -> shards) and, when the host wants them, the
gather of the controls.  Both go through torch.distributed — backend "nccl" (= RCCL over xGMI) on the GPUs,
"gloo" in the CPU tests — and never sit inside a timed tick.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced slice [lo, hi) of n instances for `rank` (the first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_rows(full, n, width, world, rank, device, dist=None, dtype=None):
    """Rank 0 holds `full` [n, width] (numpy); every rank returns its own rows as a torch tensor on `device`.
    Uneven shards are padded to the largest one for the collective and trimmed afterwards."""
    import torch
    dtype = dtype or torch.float64
    lo, hi = shard_bounds(n, world, rank)
    if world == 1:
        return torch.as_tensor(np.ascontiguousarray(full[lo:hi]), dtype=dtype).to(device)
    pad = shard_bounds(n, world, 0)[1]  # rank 0 always has the largest shard
    out = torch.empty(pad, max(width, 1), dtype=dtype, device=device)
    chunks = None
    if rank == 0:
        chunks = []
        for r in range(world):
            a, b = shard_bounds(n, world, r)
            c = torch.zeros(pad, max(width, 1), dtype=dtype)
            if width:
                c[: b - a] = torch.as_tensor(np.ascontiguousarray(full[a:b]), dtype=dtype)
            chunks.append(c.to(device))
    dist.scatter(out, chunks, src=0)
    return out[: hi - lo, :width].contiguous()


def gather_rows(local, n, world, rank, dist=None):
    """Inverse of scatter_rows: rank 0 returns the [n, width] numpy array, the other ranks None."""
    import torch
    if world == 1:
        return local.detach().cpu().numpy()
    width = local.shape[1]
    pad = shard_bounds(n, world, 0)[1]
    buf = torch.zeros(pad, width, dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0)
    if rank != 0:
        return None
    rows = []
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        rows.append(parts[r][: b - a].cpu().numpy())
    return np.concatenate(rows, axis=0)
