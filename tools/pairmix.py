#!/usr/bin/env python3
"""Do two full-chip kernels of different kinds share the SIMDs, or do they take turns?  Stream A: Merkle commits of a
2^19 x 135 matrix (VALU-bound hashing); stream B: iNTT + LDE of 135 x 2^16 polynomials (LDS / barrier / memory waits,
57 % VALU-busy alone).  Time of each alone and of both together."""
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
log_n, w = 16, 135; n = 1 << log_n; big = 8 * n
NA, NB = 2, 2          # streams per kind
sa = [torch.cuda.Stream() for _ in range(NA)]; sb = [torch.cuda.Stream() for _ in range(NB)]
A = [(torch.from_numpy(splitmix_field(big * w, seed=3 + i).view(np.int64)).to(dev), torch.zeros(p25.merkle_tree_words(big, 4), dtype=torch.int64, device=dev)) for i in range(NA)]
B = [(torch.from_numpy(splitmix_field(n * w, seed=7 + i).view(np.int64)).to(dev), torch.zeros(n * w, dtype=torch.int64, device=dev), torch.zeros(n * w, dtype=torch.int64, device=dev),
      torch.zeros(big * w, dtype=torch.int64, device=dev)) for i in range(NB)]
RA, RB = 8, 24
def run(doA, doB):
    torch.cuda.synchronize(); t = time.perf_counter()
    for r in range(max(RA, RB)):
        if doA and r < RA:
            for i in range(NA):
                check(lib.p25_merkle_commit_dev(A[i][0].data_ptr(), big, big, w, 4, A[i][1].data_ptr(), sa[i].cuda_stream))
        if doB and r < RB:
            for i in range(NB):
                check(lib.p25_lde_commit_dev(B[i][0].data_ptr(), log_n, w, 0, 3, 4, B[i][1].data_ptr(), B[i][2].data_ptr(), B[i][3].data_ptr(), None, sb[i].cuda_stream))
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
run(True, True)
ta, tb, tab = run(True, False), run(False, True), run(True, True)
print(f"hash alone {ta:.1f} ms   NTT alone {tb:.1f} ms   together {tab:.1f} ms   (sum {ta+tb:.1f}; perfect overlap would be ~{max(ta,tb):.1f} + the VALU work of the shorter)")
