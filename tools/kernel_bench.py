#!/usr/bin/env python3
"""Kernel micro-benchmarks on the synthetic matrices of SURVEY.md 8(d): SplitMix64(0x243F6A8885A308D3)
field elements, shapes 135 x 2^16 (iNTT + LDE + commit) and 2^19 x {135, 20, 16} (Merkle), through the
device-resident C-ABI primitives; reports time and algorithmic GB/s against the 8 TB/s HBM peak."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import __graft_entry__ as ge
from oracle_binding import splitmix_field
p25 = ge.load_package()
if "--lib" in sys.argv:   # A/B builds only (tools/variants.sh): an explicit path, never an environment variable
    sys.modules["plonky25_amd.binding"].lib_path = sys.argv[sys.argv.index("--lib") + 1]
p25.device_init(0)
lib = p25.lib()
from plonky25_amd.binding import _check as check
dev = torch.device("cuda", 0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def dbuf(words):
    return torch.zeros(words, dtype=torch.int64, device=dev)


rows = []
# Merkle commit of a 2^19-leaf column-major matrix
n = 1 << 19
for w in (135, 20, 16):
    host = splitmix_field(n * w, seed=0x243F6A8885A308D3).view(np.int64)
    d = torch.from_numpy(host).to(dev)
    tree = dbuf(p25.merkle_tree_words(n, 4))
    t = timeit(lambda: check(lib.p25_merkle_commit_dev(d.data_ptr(), n, n, w, 4, tree.data_ptr(), None)))
    alg = n * w * 8 + 2 * n * 32
    rows.append((f"merkle 2^19 x {w}", t, alg))
# iNTT + LDE (rate 8) + commit of 135 polynomials of 2^16 values
log_n, npoly = 16, 135
nn = 1 << log_n
host = splitmix_field(nn * npoly, seed=0x243F6A8885A308D3).view(np.int64)
d = torch.from_numpy(host).to(dev)
coeffs, tmp, lde = dbuf(nn * npoly), dbuf(nn * npoly), dbuf(8 * nn * npoly)
tree = dbuf(p25.merkle_tree_words(8 * nn, 4))
t = timeit(lambda: check(lib.p25_lde_commit_dev(d.data_ptr(), log_n, npoly, 0, 3, 4, coeffs.data_ptr(), tmp.data_ptr(),
                                                      lde.data_ptr(), tree.data_ptr(), None)))
alg = npoly * (2 * nn + 8 * nn) * 8 + 8 * nn * npoly * 8 + 2 * 8 * nn * 32
rows.append(("iNTT + LDE + commit 135 x 2^16", t, alg))
tn = timeit(lambda: check(lib.p25_lde_commit_dev(d.data_ptr(), log_n, npoly, 0, 3, 4, coeffs.data_ptr(), tmp.data_ptr(),
                                                       lde.data_ptr(), None, None)))
rows.append(("iNTT + LDE only 135 x 2^16", tn, npoly * (2 * nn + 8 * nn) * 8))
for name, t, alg in rows:
    print(f"{name:34s} {t * 1e3:8.3f} ms   {alg / 1e6:8.1f} MB algorithmic   {alg / t / 1e9:7.1f} GB/s  ({alg / t / 8e12 * 100:4.1f} % of 8 TB/s)")
