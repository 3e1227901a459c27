#!/usr/bin/env python3
"""Parity of a VARIANT library (tools/variants.sh / knobs_build.sh) against the oracle before its speed is looked at:
quotient stage on a 2^10-row circuit (satisfied and perturbed witness) and one whole fib-64 proof, byte for byte.
usage: check_variant.py path/to/libp25_x.so"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from oracle_binding import Oracle, splitmix_field
p25 = ge.load_package()
sys.modules["plonky25_amd.binding"].lib_path = sys.argv[1]
p25.device_init(0)
ora = Oracle()
inp, cfg = p25.p3_prove_fibonacci(3, 3, 4)
c = p25.Circuit.build_p3_verifier(cfg)
oc = ora.load_circuit(c.to_blob())
wires, st, msg = oc.witness(inp, seed=3)
assert st == 0, msg
for seed in (11, 12):
    b, g, a = splitmix_field(6, seed=seed).reshape(3, 2)
    zs = oc.partial_products(wires, b, g)
    assert (c.quotient(wires, zs, b, g, a) == oc.quotient(wires, zs, b, g, a)).all(), "quotient (satisfied witness)"
    bad = wires.copy()
    rng = np.random.default_rng(seed)
    for _ in range(40):
        bad[int(rng.integers(0, 135)), int(rng.integers(0, wires.shape[1]))] = int(rng.integers(0, 1 << 62))
    assert (c.quotient(bad, zs, b, g, a) == oc.quotient(bad, zs, b, g, a)).all(), "quotient (perturbed witness)"
import p3json
finp, _ = p3json.load(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json"))
fc = p25.Circuit.build_p3_verifier(p25.P3Config.fib64())
foc = ora.load_circuit(fc.to_blob())
pg, stg = fc.prove(np.stack([finp, finp]), seeds=[5, 6])
po, sto, _t, msg = foc.prove(finp, seed=5)
assert stg.tolist() == [0, 0] and sto == 0 and (pg[0] == po).all(), "fib-64 proof bytes"
print("VARIANT OK", sys.argv[1])
