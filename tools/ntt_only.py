#!/usr/bin/env python3
"""iNTT + LDE (rate 8) of 135 polynomials of 2^16 values through p25_lde_commit_dev with no tree -- the NTT launches of one
wires commitment and nothing else; for rocprofv3 passes and A/B timing of k_ntt_tile builds.
usage: ntt_only.py [--lib path/to/libp25_x.so] [--reps 8] [--log-n 16] [--polys 135]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import __graft_entry__ as ge
from oracle_binding import splitmix_field
p25 = ge.load_package()
a = sys.argv[1:]
def opt(name, default):
    return a[a.index(name) + 1] if name in a else default
if "--lib" in a:   # A/B builds only: an explicit path, never an environment variable
    sys.modules["plonky25_amd.binding"].lib_path = opt("--lib", None)
reps, log_n, npoly = int(opt("--reps", 8)), int(opt("--log-n", 16)), int(opt("--polys", 135))
p25.device_init(0)
lib = p25.lib()
from plonky25_amd.binding import _check as check
dev = torch.device("cuda", 0)
nn = 1 << log_n
d = torch.from_numpy(splitmix_field(nn * npoly, seed=0x243F6A8885A308D3).view(np.int64)).to(dev)
z = lambda w: torch.zeros(w, dtype=torch.int64, device=dev)
coeffs, tmp, lde = z(nn * npoly), z(nn * npoly), z(8 * nn * npoly)
run = lambda: check(lib.p25_lde_commit_dev(d.data_ptr(), log_n, npoly, 0, 3, 4, coeffs.data_ptr(), tmp.data_ptr(), lde.data_ptr(), None, None))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / reps
print(f"iNTT + LDE {npoly} x 2^{log_n}: {t * 1e3:.3f} ms per call ({reps} calls back to back); checksum {int(lde.sum().item()) & 0xFFFFFFFF:08x} {int(coeffs.sum().item()) & 0xFFFFFFFF:08x}")
