"""CPU: public inputs (upstream `builder.register_public_input(s)`, ProofWithPublicInputs::public_inputs) through the
builder, the blob, the proof formats and the recursive verifier, checked by the oracle's prover / verifier.

The reference's own circuit registers none (`src/p3/mod.rs:264` prints `[]`), so these circuits are the gadget of
include/p25.h kind 11 and the aggregation circuits built on the recursive verifier
(`eval_unfiltered_circuit` of the PublicInputGate: wire_i - public_inputs_hash_i, recursion.cpp)."""
import json

import numpy as np
import pytest

from conftest import P


def pi_gadget(p25, oracle, n=3):
    c = p25.Circuit.build_gadget(11, n)
    oc = oracle.load_circuit(c.to_blob())
    xs = np.array([(0x9E3779B97F4A7C15 * (i + 1)) % P for i in range(n)], dtype=np.uint64)
    expect = [int(x) for x in xs]
    acc = int(xs[0])
    for i in range(1, n):
        acc = acc * int(xs[i]) % P
        expect.append(acc)
    return c, oc, xs, np.array(expect, dtype=np.uint64)


def test_registered_public_inputs_reach_the_proof(p25, oracle):
    c, oc, xs, expect = pi_gadget(p25, oracle)
    assert int(c.info.num_public_inputs) == 5 and "PoseidonGate" in " ".join(c.gate_counts())
    proof, st, _t, msg = oc.prove(xs, seed=3)
    assert st == 0, msg
    assert proof.size == int(c.info.proof_words) and (c.public_inputs(proof) == expect).all()
    dg, cap = oc.digest()
    assert oc.verify(proof, dg, cap)[0] == 0
    # the hash of the public inputs is what the PublicInputGate row holds, and every constraint of the circuit holds
    wires, st, msg = oc.witness(xs, seed=3)
    pi_rows = np.nonzero((wires[:4] == oracle.hash_no_pad(expect)[:, None]).all(axis=0))[0]
    assert st == 0 and pi_rows.size >= 1 and oc.check_constraints(wires)[0] == 0
    # a proof whose public inputs were altered afterwards is rejected (they enter the transcript through their hash)
    for k in range(1, 6):
        bad = proof.copy()
        bad[-k] = (int(bad[-k]) + 1) % P
        assert oc.verify(bad, dg, cap)[0] != 0
    # formats: serde JSON and upstream's binary form carry them
    js = json.loads(c.proof_to_json(proof))
    assert js["public_inputs"] == [int(v) for v in expect]
    raw = c.proof_to_bytes(proof)
    assert raw[-40:] == expect.astype("<u8").tobytes() and (c.proof_from_bytes(raw) == proof).all()


def test_circuit_without_public_inputs_is_unchanged(p25, oracle):
    """Same rows, same digest inputs as before the feature: hash_no_pad([]) is four zeros and costs no gate."""
    c = p25.Circuit.build_gadget(0, 0)
    assert int(c.info.num_public_inputs) == 0 and not any(k.startswith("PoseidonGate") for k in c.gate_counts())
    assert c.public_inputs(np.zeros(int(c.info.proof_words), dtype=np.uint64)).size == 0


def test_aggregator_exposes_a_commitment_to_what_it_verified(p25, oracle):
    """Two proofs of a circuit WITH public inputs -> aggregator: its 4 public inputs are hash_no_pad(pi_0 || pi_1);
    one more level on top (aggregates have public inputs themselves): hash_no_pad(root_a || root_b).  A leaf circuit
    WITHOUT public inputs is identified by hash_no_pad(its wires cap)."""
    c, oc, xs, expect = pi_gadget(p25, oracle, n=2)
    dg, cap = oc.digest()
    proofs = []
    for s in range(4):
        x = (xs + np.uint64(s)) % np.uint64(P)
        pr, st, _t, msg = oc.prove(x, seed=s)
        assert st == 0, msg
        proofs.append(pr)
    agg = c.build_aggregator(2, digest=dg, cs_cap=cap)
    assert int(agg.info.num_public_inputs) == 4
    oa = oracle.load_circuit(agg.to_blob())
    roots, agg_proofs = [], []
    for k in range(2):
        both = np.concatenate([proofs[2 * k], proofs[2 * k + 1]])
        ap, st, _t, msg = oa.prove(both, seed=10 + k)
        assert st == 0, msg
        assert oa.verify(ap)[0] == 0
        want = oracle.hash_no_pad(np.concatenate([c.public_inputs(proofs[2 * k]), c.public_inputs(proofs[2 * k + 1])]))
        assert (agg.public_inputs(ap) == want).all()
        roots.append(want)
        agg_proofs.append(ap)
    # an inner proof with altered public inputs has no witness in the aggregator (its hash feeds the inner transcript)
    bad = np.concatenate([proofs[0], proofs[1]])
    bad[proofs[0].size - 1] = (int(bad[proofs[0].size - 1]) + 1) % P
    assert oa.witness(bad, seed=1)[1] == 4
    # second level
    adg, acap = oa.digest()
    top = agg.build_aggregator(2, digest=adg, cs_cap=acap)
    ot = oracle.load_circuit(top.to_blob())
    w, st, msg = ot.witness(np.concatenate(agg_proofs), seed=5)
    assert st == 0, msg
    assert ot.check_constraints(w)[0] == 0
    # leaves without public inputs: identified by the hash of their wires cap
    leaf = p25.Circuit.build_gadget(8, 0)
    ol = oracle.load_circuit(leaf.to_blob())
    lp, st, _t, msg = ol.prove(np.array([7, 7, 49], dtype=np.uint64), seed=1)
    assert st == 0, msg
    ldg, lcap = ol.digest()
    la = leaf.build_aggregator(1, digest=ldg, cs_cap=lcap)
    ola = oracle.load_circuit(la.to_blob())
    ap, st, _t, msg = ola.prove(lp, seed=2)
    assert st == 0, msg
    assert (la.public_inputs(ap) == oracle.hash_no_pad(lp[:64])).all() and ola.verify(ap)[0] == 0
