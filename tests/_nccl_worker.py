"""Worker for tests/test_gpu_nccl.py: a WORLD-SIZE-1 `nccl` (= RCCL) process group on cuda:0, so that everything the
8-GPU bench does through RCCL -- init with `device_id=`, the gather of device-resident proofs and statuses of the real
dtypes and shapes, the int64 broadcast, the float64 MAX all-reduce and all-gather, the int32 MIN all-reduce, barriers --
has executed on RCCL before the driver's scaling run.  One process, one GPU; started as a child of the test."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    ge.load_package()
    from plonky25_amd import dist as pd
    # the gather of one bench step: 256 proofs x 19,861 words int64 + int32 statuses, on device tensors, twice (two
    # steps in flight, the slots of bench.py), on a side stream like bench.py's
    n, words = 256, 19861
    g = pd.ProofGatherer(n, words, dev, slots=2)
    side = torch.cuda.Stream(device=dev)
    base = torch.arange(n * words, dtype=torch.int64, device=dev).reshape(n, words)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    st[7] = 4
    for step in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            blocks, sts = g.gather(base + step, st, slot=step & 1)
    torch.cuda.synchronize()
    assert len(blocks) == 1 and torch.equal(blocks[0], base + 2) and torch.equal(sts[0], st)
    # uneven form (padded staging) cannot occur with one rank; the one-shot form
    gp, gs = pd.gather_proofs(base[:5], st[:5], 5)
    assert torch.equal(gp, base[:5]) and gp.device.type == "cuda"
    # plonky3 input variants: int64 broadcast of uint64 data with the top bit set
    src = np.arange(24, dtype=np.uint64).reshape(4, 6) * np.uint64(0xF000000000000001)
    got = pd.broadcast_int64(src, (4, 6), dev)
    assert got.dtype == np.uint64 and (got == src).all()
    # timing reductions: float64 MAX all-reduce, float64 all-gather; status agreement: int32 MIN / MAX
    assert pd.max_over_ranks(3.25, dev) == 3.25
    mine = torch.tensor([1.5, 0.25, 256.0], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(1)]
    dist.all_gather(allr, mine)
    assert torch.equal(allr[0], mine)
    for op, v in ((dist.ReduceOp.MIN, 1), (dist.ReduceOp.MAX, 0)):
        t = torch.tensor([v], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=op)
        assert int(t.item()) == v
    dist.barrier()
    torch.cuda.synchronize()
    print("NCCL_WORLD1_OK", torch.cuda.get_device_name(0))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
