#!/usr/bin/env python3
"""Prove the artifact once on the GPU (after one warm-up) -- target for rocprofv3 traces."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
p25 = ge.load_package()
if "--lib" in sys.argv:   # profiling builds only (tools/qmask.sh): an explicit path, never an environment variable
    i = sys.argv.index("--lib")
    sys.modules["plonky25_amd.binding"].lib_path = sys.argv[i + 1]
    del sys.argv[i:i + 2]
late = p25.device_init(0)
print("runtime:", p25.runtime_info().as_dict(), "hw-queue request late:", late)   # under rocprofv3 the profiler opens the GPU first
if "--log-n" in sys.argv:   # another inner trace height (BASELINE config 5: --log-n 20), from the native plonky3 prover
    i = sys.argv.index("--log-n")
    inputs, cfg = p25.p3_prove_fibonacci(int(sys.argv[i + 1]), 100, 16, threads=os.cpu_count() or 1)
    del sys.argv[i:i + 2]
    c = p25.Circuit.build_p3_verifier(cfg)
else:
    inputs, _ = p25.p3_proof_from_json(open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")).read())
    c = p25.Circuit.build_p3_verifier(p25.P3Config.fib64())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c.prove(inputs, seeds=[0])
if n == 1:  # a lone proof: per-phase device times (and the latency-oriented kernel forms)
    proofs, st, tm = c.prove(inputs, seeds=[0], timings=True)
    print(st.tolist(), {k: round(v, 3) for k, v in tm.as_dict().items()})
else:       # a small batch: the throughput path's kernels, several proofs in flight
    proofs, st = c.prove(np.stack([inputs] * n), seeds=list(range(n)))
    print(st.tolist())
