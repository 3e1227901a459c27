#!/usr/bin/env python3
"""BASELINE.json config 5: inner Fibonacci STARK with 2^log_n rows (default 20) -> verifier circuit of
2^19 rows, LDE 2^22.  Proves it on the GPU, checks the proof with the oracle verifier and (optionally)
compares bytes with the oracle prover."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
from oracle_binding import Oracle

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
full_parity = "--parity" in sys.argv
p25 = ge.load_package(); p25.device_init(0)
t = time.time(); inp, cfg = p25.p3_prove_fibonacci(log_n, 100, 16, threads=os.cpu_count()); t_p3 = time.time() - t
t = time.time(); c = p25.Circuit.build_p3_verifier(cfg); info = c.info; t_build = time.time() - t
t = time.time(); dg, cap = c.digest(); t_dev = time.time() - t
proofs, st, tm = c.prove(inp, seeds=[5], timings=True)
t = time.time(); proofs2, st2 = c.prove(np.stack([inp] * 4), seeds=[5, 6, 7, 8]); t4 = time.time() - t
ora = Oracle(); oc = ora.load_circuit(c.to_blob())
code, msg = oc.verify(proofs[0], dg, cap)
out = {"log_n": log_n, "inputs": int(len(inp)), "p3_prove_s": round(t_p3, 2), "circuit_build_s": round(t_build, 2),
       "device_tables_s": round(t_dev, 2), "rows_used": int(info.num_rows_used), "degree_bits": int(info.degree_bits),
       "proof_words": int(info.proof_words), "status": st.tolist(), "status4": st2.tolist(),
       "gpu_phase_ms": {k: round(v, 2) for k, v in tm.as_dict().items()}, "batch4_wall_s": round(t4, 2),
       "oracle_verifier": code, "msg": msg, "same_seed_same_bytes": bool((proofs[0] == proofs2[0]).all())}
if full_parity:
    t = time.time(); po, sto, otm, m = oc.prove(inp, seed=5); out["oracle_prove_s"] = round(time.time() - t, 1)
    out["oracle_status"] = sto; out["bytes_equal"] = bool((po == proofs[0]).all())
print(json.dumps(out))
