#!/usr/bin/env python3
"""Aggregation tree over a batch of fib-64 plonky3-verifier proofs (the north star's "final aggregation"
taken literally): N leaf proofs -> N/2 proofs of a 2-to-1 recursive verifier -> ... -> one root proof, every level a
plain batch prove on the GPU.  Prints one JSON line with the per-level circuit sizes and times; the root is checked
by the oracle's verifier.   usage: aggregate.py [N = 64, a power of two]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
assert n >= 2 and n & (n - 1) == 0
p25 = ge.load_package(); p25.device_init(0)
with open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")) as f:
    base, cfg = p25.p3_proof_from_json(f.read())
variants = [base] + [p25.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)[0] for v in range(1, 8)]
circ = p25.Circuit.build_p3_verifier(cfg)
circ.digest()
batch = np.stack([variants[i % 8] for i in range(n)])
circ.prove(batch[:16], seeds=list(range(16)))                      # warm-up: contexts, tables
t = time.perf_counter(); level, st = circ.prove(batch, seeds=np.arange(n, dtype=np.uint64)); dt = time.perf_counter() - t
assert (st == 0).all()
levels = [{"level": 0, "circuit_rows_log2": int(circ.info.degree_bits), "proofs": n, "prove_s": round(dt, 3)}]
total = dt
while len(level) > 1:
    t = time.perf_counter(); circ = circ.build_aggregator(2); circ.digest(); build = time.perf_counter() - t
    pairs = np.stack([np.concatenate([level[2 * i], level[2 * i + 1]]) for i in range(len(level) // 2)])
    circ.prove(pairs[:min(16, len(pairs))], seeds=list(range(min(16, len(pairs)))))   # warm-up: this circuit's contexts
    t = time.perf_counter(); level, st = circ.prove(pairs, seeds=np.arange(len(pairs), dtype=np.uint64)); dt = time.perf_counter() - t
    assert (st == 0).all(), st
    total += dt
    levels.append({"level": len(levels), "circuit_rows_log2": int(circ.info.degree_bits), "rows_used": int(circ.info.num_rows_used),
                   "proofs": len(level), "prove_s": round(dt, 3), "circuit_build_s": round(build, 2)})
from oracle_binding import Oracle
oc = Oracle().load_circuit(circ.to_blob())
dg, cap = circ.digest()
code, msg = oc.verify(level[0], dg, cap)
print(json.dumps({"leaf_proofs": n, "levels": levels, "gpu_prove_s_total": round(total, 3),
                  "leaf_equivalent_proofs_per_s_including_aggregation": round(n / total, 2),
                  "root_proof_words": int(level[0].size), "oracle_verifier_accepts_root": code == 0, "msg": msg}))
