"""Multi-GPU plumbing: proofs are independent, so a batch shards across ranks with no data-path
collective; the only exchange is the final aggregation of finished proofs (+ statuses) on rank 0
(RCCL over xGMI with backend "nccl", gloo on CPU in the tests).  One process per GPU."""
import torch
import torch.distributed as dist

__all__ = ["shard_range", "gather_proofs", "max_over_ranks"]


def shard_range(n_total, rank, world):
    """Contiguous block partition of n_total proofs: rank r gets [start, stop)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_proofs(local_proofs, local_status, n_total, dst=0):
    """Gathers per-rank proofs [n_local, words] (int64) and statuses [n_local] (int32) onto `dst` in
    global proof order.  Shards may differ in size by one (padded to the largest for the collective)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    m = max(sizes)
    words = local_proofs.shape[1]
    pad_p = torch.zeros((m, words), dtype=local_proofs.dtype, device=local_proofs.device)
    pad_s = torch.full((m,), -1, dtype=local_status.dtype, device=local_status.device)
    pad_p[: local_proofs.shape[0]] = local_proofs
    pad_s[: local_status.shape[0]] = local_status
    gp = [torch.zeros_like(pad_p) for _ in range(world)] if rank == dst else None
    gs = [torch.zeros_like(pad_s) for _ in range(world)] if rank == dst else None
    dist.gather(pad_p, gp, dst=dst)
    dist.gather(pad_s, gs, dst=dst)
    if rank != dst:
        return None, None
    proofs = torch.cat([gp[r][: sizes[r]] for r in range(world)])
    status = torch.cat([gs[r][: sizes[r]] for r in range(world)])
    return proofs, status


def max_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
