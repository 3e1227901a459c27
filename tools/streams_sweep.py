#!/usr/bin/env python3
"""proofs/s of the fib-64 batch (256 proofs per step) against the number of proofs kept in flight."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
dev = torch.device("cuda", 0)
inputs, _ = p25.p3_proof_from_json(open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")).read())
circ = p25.Circuit.build_p3_verifier(p25.P3Config.fib64()); circ.digest()
pw = int(circ.info.proof_words)
B, steps = 256, 2
d_in = torch.from_numpy(np.stack([inputs] * B).view(np.int64)).to(dev)
d_seeds = torch.arange(B, dtype=torch.int64, device=dev)
d_proofs = torch.zeros((B, pw), dtype=torch.int64, device=dev)
d_status = torch.zeros(B, dtype=torch.int32, device=dev)
for k in [int(x) for x in sys.argv[1:]] or [16, 12, 14, 18, 20, 24, 16]:
    circ.set_streams(k)
    for it in range(1 + steps):
        if it == 1:
            circ.sync(); torch.cuda.synchronize(); t = time.perf_counter()
        circ.prove_dev(d_in.data_ptr(), B, d_seeds.data_ptr(), d_proofs.data_ptr(), pw, d_status.data_ptr())
    circ.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(json.dumps({"streams": k, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "lib default"), "proofs_per_s": round(B * steps / dt, 2),
                      "ok": bool((d_status.cpu().numpy() == 0).all())}), flush=True)
