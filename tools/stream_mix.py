#!/usr/bin/env python3
"""Which kernels does a hash-kernel variant interact with?  16 streams, each issuing the Merkle commits (mode merkle)
or the iNTT + LDE + Merkle commits (mode lde) of a proof's three oracles back to back through the device-resident
C-ABI primitives; reports commits-of-a-proof per second.  (tools/hash_variants.sh builds the variants.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import __graft_entry__ as ge
from oracle_binding import splitmix_field
p25 = ge.load_package(); p25.device_init(0)
lib = p25.lib()
from plonky25_amd.binding import _check as check
dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "merkle"
S, R = 16, 6
log_n = 16; n = 1 << log_n; big = 8 * n
widths = (135, 20, 16)
streams = [torch.cuda.Stream() for _ in range(S)]
bufs = []
for s in range(S):
    per = []
    for w in widths:
        if mode == "merkle":
            d = torch.from_numpy(splitmix_field(big * w, seed=1 + s).view(np.int64)).to(dev)
            per.append((d, torch.zeros(p25.merkle_tree_words(big, 4), dtype=torch.int64, device=dev)))
        else:
            d = torch.from_numpy(splitmix_field(n * w, seed=1 + s).view(np.int64)).to(dev)
            per.append((d, torch.zeros(n * w, dtype=torch.int64, device=dev), torch.zeros(n * w, dtype=torch.int64, device=dev),
                        torch.zeros(big * w, dtype=torch.int64, device=dev), torch.zeros(p25.merkle_tree_words(big, 4), dtype=torch.int64, device=dev)))
    bufs.append(per)

def issue():
    for r in range(R):
        for s in range(S):
            st = streams[s].cuda_stream
            for w, b in zip(widths, bufs[s]):
                if mode == "merkle":
                    check(lib.p25_merkle_commit_dev(b[0].data_ptr(), big, big, w, 4, b[1].data_ptr(), st))
                else:
                    check(lib.p25_lde_commit_dev(b[0].data_ptr(), log_n, w, 0, 3, 4, b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), b[4].data_ptr(), st))
issue(); torch.cuda.synchronize()
t = time.perf_counter(); issue(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"{mode}: {S} streams x {R} rounds of (135, 20, 16)-column commits: {dt*1e3:.1f} ms -> {S*R/dt:.1f} proof-commit-sets/s")
