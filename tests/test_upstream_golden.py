"""Consumes the output of tools/upstream_check (a run of the REAL plonky2 @ 3de92d9 prover on the reference's
test circuit) when it has been dropped into tests/golden/; skipped otherwise.  See tools/upstream_check/README.md."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
need = pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "upstream_circuit.json")),
                          reason="no upstream run available (tools/upstream_check has to be run where cargo exists)")


def flatten_upstream_proof(js):
    """serde JSON of ProofWithPublicInputs -> the flat word layout of include/p25.h."""
    out = []
    pr = js["proof"]

    def cap(c):
        for h in c:
            out.extend(int(x) for x in h["elements"])

    def exts(v):
        for e in v:
            out.extend(int(x) for x in e)

    cap(pr["wires_cap"]); cap(pr["plonk_zs_partial_products_cap"]); cap(pr["quotient_polys_cap"])
    op = pr["openings"]
    for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products", "quotient_polys"):
        exts(op[k])
    fp = pr["opening_proof"]
    for c in fp["commit_phase_merkle_caps"]:
        cap(c)
    for q in fp["query_round_proofs"]:
        for leaf, path in q["initial_trees_proof"]["evals_proofs"]:
            out.extend(int(x) for x in leaf)
            cap(path["siblings"])
        for step in q["steps"]:
            exts(step["evals"])
            cap(step["merkle_proof"]["siblings"])
    exts(fp["final_poly"]["coeffs"])
    out.append(int(fp["pow_witness"]))
    return np.array(out, dtype=np.uint64)


@need
def test_circuit_shape_and_verifier_data_equal_upstream(p25, fib_circuit, fib_oracle):
    up = json.load(open(os.path.join(GOLD, "upstream_circuit.json")))
    assert int(fib_circuit.info.degree_bits) == up["degree_bits"]
    counts = fib_circuit.gate_counts()
    assert list(counts.keys()) == up["gate_ids"]
    assert list(counts.values()) == up["rows_per_gate"]
    assert int(fib_circuit.info.num_generators) == up["num_generators"]
    assert int(fib_circuit.info.num_gate_constraints) == up["num_gate_constraints"]
    dg, cap = fib_oracle.digest()
    assert [int(x) for x in dg] == up["circuit_digest"]
    assert [[int(x) for x in h] for h in cap] == up["constants_sigmas_cap"]


@need
def test_oracle_verifier_accepts_upstream_proof(fib_oracle):
    pj = os.path.join(GOLD, "upstream_proof.json")
    if not os.path.exists(pj):
        pytest.skip("no upstream proof")
    proof = flatten_upstream_proof(json.load(open(pj)))
    assert proof.size == fib_oracle.proof_words
    code, msg = fib_oracle.verify(proof)
    assert code == 0, msg


@need
@pytest.mark.gpu
def test_gpu_reproduces_upstream_proof_bytes(gpu, fib_circuit, fib_inputs):
    fj, pj = os.path.join(GOLD, "upstream_filler.json"), os.path.join(GOLD, "upstream_proof.json")
    if not (os.path.exists(fj) and os.path.exists(pj)):
        pytest.skip("no upstream filler / proof")
    filler = np.array(json.load(open(fj))["filler"], dtype=np.uint64)
    want = flatten_upstream_proof(json.load(open(pj)))
    proofs, st = fib_circuit.prove_filler(fib_inputs, filler)
    assert st.tolist() == [0]
    diff = np.nonzero(proofs[0] != want)[0]
    assert diff.size == 0, f"first differing proof words {diff[:8]}"


@need
def test_binary_proof_form_equals_upstream(p25, fib_circuit):
    """`proof.to_bytes()` of the real crate against p25_proof_to_bytes on the same proof (parsed from its JSON)."""
    pb, pj = os.path.join(GOLD, "upstream_proof.bin"), os.path.join(GOLD, "upstream_proof.json")
    if not (os.path.exists(pb) and os.path.exists(pj)):
        pytest.skip("no upstream binary proof")
    want = open(pb, "rb").read()
    flat = flatten_upstream_proof(json.load(open(pj)))
    assert fib_circuit.proof_to_bytes(flat) == want
    assert (fib_circuit.proof_from_bytes(want) == flat).all()


@need
@pytest.mark.gpu
def test_circuit_data_bytes_equal_upstream(gpu, fib_circuit):
    """`data.to_bytes(..)` of the real crate (length, SHA-256, first 64 KiB) against p25_circuit_to_bytes."""
    meta, head = os.path.join(GOLD, "upstream_circuit_data.json"), os.path.join(GOLD, "upstream_circuit_data_head.bin")
    if not (os.path.exists(meta) and os.path.exists(head)):
        pytest.skip("no upstream CircuitData bytes")
    import hashlib
    up = json.load(open(meta))
    mine = fib_circuit.to_bytes()
    h = open(head, "rb").read()
    first = next((i for i, (a, b) in enumerate(zip(mine, h)) if a != b), None)
    assert first is None, f"CircuitData bytes differ from upstream's at offset {first}"
    assert len(mine) == up["len"] and hashlib.sha256(mine).hexdigest() == up["sha256"]


need_rec = pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "upstream_recursive_circuit.json")),
                              reason="no upstream run of builder.verify_proof available (tools/upstream_check)")


@need_rec
def test_recursive_verifier_shape_vs_upstream(p25, fib_circuit, fib_oracle):
    """The day cargo exists: how far `p25_circuit_build_recursive_verifier` is from upstream's `builder.verify_proof`
    circuit for one fib-64 proof.  The gate SET must agree (the library claims upstream's gate set); rows per gate and
    the digest are reported, and asserted only once DESIGN.md section 7 stops disclaiming row-for-row equality."""
    up = json.load(open(os.path.join(GOLD, "upstream_recursive_circuit.json")))
    dg, cap = fib_oracle.digest()
    rc = fib_circuit.build_recursive_verifier(1, dg, cap)
    counts = rc.gate_counts()
    assert sorted(counts.keys()) == sorted(up["gate_ids"]), (sorted(counts.keys()), sorted(up["gate_ids"]))
    print("rows per gate (libp25 / upstream):", {k: (counts[k], dict(zip(up["gate_ids"], up["rows_per_gate"]))[k]) for k in counts})
    print("degree bits (libp25 / upstream):", int(rc.info.degree_bits), up["degree_bits"])
