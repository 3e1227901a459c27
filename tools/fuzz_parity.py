#!/usr/bin/env python3
"""Randomised differential run: GPU proofs against the oracle's, byte for byte, over circuits and inputs the fixed tests do
not visit -- seeded AIR families (linear recurrences of random width, quadratic pairs, the degree-3 AIRs, the degree-4 / 5 families with FriConfig.log_blowup 2 / 3) at random trace
heights / query counts / PoW bits, the gadget circuits on random operands, the reference-gates circuit (gadget 14).
usage: fuzz_parity.py [seconds=240] [seed=1]      prints one line per case and a summary; exit code 1 on any difference."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from oracle_binding import Oracle
import air_cases
from gadget_cases import cases as gadget_cases
from test_blob_independent_writer import reference_gates_inputs
P = 0xFFFFFFFF00000001
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
p25 = ge.load_package()
p25.device_init(0)
ora = Oracle()
t0, n_ok, n_bad = time.time(), 0, 0


def check(name, circ, inputs, seeds):
    global n_ok, n_bad
    oc = ora.load_circuit(circ.to_blob())
    proofs, st = circ.prove(np.stack(inputs), seeds=seeds)
    for i, (inp, sd) in enumerate(zip(inputs, seeds)):
        po, sto, _t, msg = oc.prove(inp, seed=sd)
        same = int(st[i]) == sto and (sto != 0 or (proofs[i] == po).all())
        n_ok += same
        n_bad += not same
        if not same:
            print(f"DIFF {name} case {i}: gpu status {int(st[i])} oracle {sto} {msg}", flush=True)
    print(f"{time.time() - t0:6.1f}s {name}: {len(inputs)} proofs, rows 2^{int(circ.info.degree_bits)}", flush=True)


while time.time() - t0 < budget:
    kind = int(rng.integers(0, 6))
    seed = int(rng.integers(1, 1 << 30))
    if kind == 0:
        width = int(rng.integers(2, 10))
        air, coef = air_cases.random_recurrence(p25, seed, width)
        log_n = int(rng.integers(2, 7))
        trace = air_cases.random_recurrence_trace(coef, log_n)
        name = f"recurrence w{width} 2^{log_n}"
    elif kind == 1:
        air, par = air_cases.quadratic_pair(p25, seed)
        log_n = int(rng.integers(2, 7))
        trace = air_cases.quadratic_pair_trace(par, log_n)
        name = f"quadratic_pair 2^{log_n}"
    elif kind == 2:
        which = "cubic" if seed & 1 else "cubic_transition"
        air = getattr(air_cases, which)(p25)
        log_n = int(rng.integers(2, 7))
        trace = getattr(air_cases, which + "_trace")(log_n)
        name = f"{which} 2^{log_n}"
    elif kind == 5:      # degree 4 / 5: four quotient chunks, FriConfig.log_blowup 2 or 3 (round 6)
        which = "quartic_map" if seed & 1 else "quintic_selector"
        air, par = getattr(air_cases, which)(p25, seed)
        log_n = int(rng.integers(2, 6))
        trace = getattr(air_cases, which + "_trace")(par, log_n)
        blowup = 2 + int(seed >> 1 & 1)
        name = f"{which} 2^{log_n} log_blowup {blowup}"
    elif kind == 3:
        c = p25.Circuit.build_gadget(14, 0)
        ins = [reference_gates_inputs(ora, *(int(v) for v in rng.integers(0, 1 << 32, size=3))) for _ in range(3)]
        check("reference gates", c, ins, [seed, seed + 1, seed + 2])
        continue
    else:
        allc = gadget_cases(ora)
        nm, gk, param, vals = allc[int(rng.integers(0, len(allc)))]
        c = p25.Circuit.build_gadget(gk, param)
        check(f"gadget {nm}", c, [np.array(vals, dtype=np.uint64)], [seed])
        continue
    q, pw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    lb = blowup if kind == 5 else 1
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=q, pow_bits=pw, log_blowup=lb)
    alt, _ = p25.p3_prove_air(air, trace, num_queries=q, pow_bits=pw, pow_start=1 << 20, log_blowup=lb)
    circ = p25.Circuit.build_p3_verifier_air(cfg, air)
    bad = inp.copy()
    k = int(rng.integers(0, inp.size))
    bad[k] = (int(bad[k]) + 1) % P                                   # both sides must fail it alike
    check(name + f" q{q} pow{pw}", circ, [inp, alt, bad], [seed, seed + 1, seed + 2])
print(f"FUZZ {'OK' if n_bad == 0 else 'FAILED'}: {n_ok} agreeing proofs, {n_bad} differences, {time.time() - t0:.0f} s")
sys.exit(1 if n_bad else 0)
