#!/usr/bin/env python3
"""BASELINE config 5 (outer circuit 2^19 rows): proofs/s against the batch size per step -- does the step end in a
ragged round of the proving streams?  (bench.py's configs block uses one of these sizes.)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
dev = torch.device("cuda", 0)
inp, cfg = p25.p3_prove_fibonacci(20, 100, 16, threads=os.cpu_count())
circ = p25.Circuit.build_p3_verifier(cfg); circ.digest()
pw = int(circ.info.proof_words)
free0 = torch.cuda.mem_get_info()[0]
for B, steps in ((16, 2), (13, 2), (26, 2), (32, 1), (16, 4), (20, 2)):
    d_in = torch.from_numpy(np.stack([inp] * B).view(np.int64)).to(dev)
    d_seeds = torch.arange(B, dtype=torch.int64, device=dev)
    d_proofs = torch.zeros((B, pw), dtype=torch.int64, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    for it in range(1 + steps):
        if it == 1:
            torch.cuda.synchronize(); t = time.perf_counter()
        circ.prove_dev(d_in.data_ptr(), B, d_seeds.data_ptr(), d_proofs.data_ptr(), pw, d_status.data_ptr())
        circ.sync()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    free1 = torch.cuda.mem_get_info()[0]
    print(json.dumps({"batch": B, "steps": steps, "proofs_per_s": round(B * steps / dt, 3), "ok": bool((d_status.cpu().numpy() == 0).all()),
                      "device_GB_in_use_by_contexts": round((free0 - free1) / 1e9, 1)}), flush=True)
