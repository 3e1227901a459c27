#!/usr/bin/env python3
"""Aggregation tree over a batch of fib-64 plonky3-verifier proofs (the north star's "final aggregation"
taken literally): N leaf proofs folded `arity` at a time (plonky25_amd.aggregate.fold, the code bench.py runs on every
rank) down to one root proof, every level a plain batch prove on the GPU.  Prints one JSON line with the per-level
circuit sizes and times; the root is checked by the oracle's verifier.  The profiling target for the aggregator's
kernels: `rocprofv3 --kernel-trace --stats -- python3 tools/aggregate.py 64 8`.
usage: aggregate.py [N = 64] [arity = 8; bench.py's default is 13] [--no-leaves: read nothing, prove the leaves untimed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 64
arity = int(args[1]) if len(args) > 1 else 8
assert n >= 2 and n & (n - 1) == 0
p25 = ge.load_package(); p25.device_init(0)
from plonky25_amd import aggregate as ag
with open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")) as f:
    base, cfg = p25.p3_proof_from_json(f.read())
variants = [base] + [p25.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)[0] for v in range(1, 8)]
circ = p25.Circuit.build_p3_verifier(cfg)
circ.digest()
batch = np.stack([variants[i % 8] for i in range(n)])
circ.prove(batch[:16], seeds=list(range(16)))                      # warm-up: contexts, tables
t = time.perf_counter(); level, st = circ.prove(batch, seeds=np.arange(n, dtype=np.uint64)); dt = time.perf_counter() - t
assert (st == 0).all()
f = ag.fold(circ, [level[i] for i in range(n)], arity=arity)
from oracle_binding import Oracle
ora = Oracle()
oc = ora.load_circuit(f["top"].to_blob())
dg, cap = f["top"].digest()
code, msg = oc.verify(f["root"], dg, cap)
want = ag.expected_commitment([level[i][:ag.CAP_WORDS] for i in range(n)], arity, ora.hash_no_pad)
total = dt + f["tree_s"]
print(json.dumps({"leaf_proofs": n, "arity": arity, "leaf_prove_s": round(dt, 4), "levels": f["levels"],
                  "tree_prove_s": round(f["tree_s"], 4), "gpu_prove_s_total": round(total, 4),
                  "leaf_equivalent_proofs_per_s_including_aggregation": round(n / total, 2),
                  "root_proof_words": int(f["root"].size), "oracle_verifier_accepts_root": code == 0,
                  "root_public_inputs_commit_to_the_leaves": [int(v) for v in f["top"].public_inputs(f["root"])] == want,
                  "msg": msg}))
