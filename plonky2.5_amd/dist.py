"""Multi-GPU plumbing: proofs are independent, so a batch shards across ranks with no data-path
collective; the only exchange is the final aggregation of finished proofs (+ statuses) on rank 0
(RCCL over xGMI with backend "nccl", gloo on CPU in the tests).  One process per GPU."""
import torch
import torch.distributed as dist

__all__ = ["shard_range", "shard_sizes", "gather_proofs", "ProofGatherer", "max_over_ranks", "broadcast_int64"]


def shard_range(n_total, rank, world):
    """Contiguous block partition of n_total proofs: rank r gets [start, stop)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


class ProofGatherer:
    """The final aggregation step with every buffer allocated ONCE: per-rank proofs [n_local, words] (int64) and
    statuses [n_local] (int32) are gathered onto `dst` in global proof order.  Shards may differ in size by one; the
    collective runs on blocks padded to the largest shard.  `device` is where the collective's tensors live (the
    rank's GPU for RCCL, "cpu" for gloo).  `slots`: gathers of consecutive steps are in flight together when the steps
    are pipelined (bench.py), so the padded staging blocks exist once per slot; the receive buffers are shared (the
    collectives of one process group execute in issue order) and always hold the LATEST gather."""

    def __init__(self, n_total, words, device, dst=0, proof_dtype=torch.int64, status_dtype=torch.int32, slots=1):
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), dst
        self.n_total, self.words = n_total, words
        self.sizes = shard_sizes(n_total, self.world)
        self.m = max(self.sizes)
        self.n_local = self.sizes[self.rank]
        self.even = all(s == self.m for s in self.sizes)
        # send side: the caller's own tensors when every shard is full-size, a padded staging block otherwise
        self.pad_p = None if self.even else [torch.zeros((self.m, words), dtype=proof_dtype, device=device)
                                             for _ in range(slots)]
        self.pad_s = None if self.even else [torch.full((self.m,), -1, dtype=status_dtype, device=device)
                                             for _ in range(slots)]
        if self.rank == dst:
            self.gp = [torch.zeros((self.m, words), dtype=proof_dtype, device=device) for _ in range(self.world)]
            self.gs = [torch.zeros((self.m,), dtype=status_dtype, device=device) for _ in range(self.world)]
        else:
            self.gp = self.gs = None

    def gather(self, local_proofs, local_status, slot=0):
        """Returns (list of per-rank proof blocks, list of per-rank status blocks) on dst -- views of the
        pre-allocated receive buffers trimmed to each shard's size -- and (None, None) elsewhere.  The collectives are
        issued on the CURRENT stream's order (torch's process group makes its own stream wait for it and the current
        stream wait for the result), so a caller on a side stream does not block the host."""
        assert local_proofs.shape == (self.n_local, self.words), (local_proofs.shape, self.n_local, self.words)
        if self.even:
            sp, ss = local_proofs.contiguous(), local_status.contiguous()
        else:
            sp, ss = self.pad_p[slot], self.pad_s[slot]
            sp[: self.n_local].copy_(local_proofs)
            ss[: self.n_local].copy_(local_status)
        dist.gather(sp, self.gp, dst=self.dst)
        dist.gather(ss, self.gs, dst=self.dst)
        if self.rank != self.dst:
            return None, None
        return ([self.gp[r][: self.sizes[r]] for r in range(self.world)],
                [self.gs[r][: self.sizes[r]] for r in range(self.world)])


def gather_proofs(local_proofs, local_status, n_total, dst=0):
    """One-shot form of ProofGatherer: gathers onto `dst` and concatenates in global proof order."""
    g = ProofGatherer(n_total, local_proofs.shape[1], local_proofs.device, dst, local_proofs.dtype, local_status.dtype)
    gp, gs = g.gather(local_proofs, local_status)
    if gp is None:
        return None, None
    return torch.cat(gp), torch.cat(gs)


def broadcast_int64(array_or_none, shape, device, src=0):
    """Broadcasts an int64 array of known shape from `src` (numpy in, numpy out): the plonky3 input variants are
    generated once, on rank 0, and shipped to the other ranks."""
    import numpy as np
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(array_or_none).view(np.int64)).reshape(shape).to(device)
    else:
        t = torch.zeros(shape, dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().view(np.uint64)


def max_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
