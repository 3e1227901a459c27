"""CPU: recursion at the gate level and as a whole circuit, checked by the oracle.

1. In the spirit of the reference's `test_eval_fns` (/root/reference/src/common/poseidon2/poseidon2_gate.rs:575-581,
   arithmetic_u32.rs:494-499 ...): for every gate of the inner circuits, the base-field evaluator, the extension
   evaluator and the IN-CIRCUIT evaluator (`eval_unfiltered_circuit`, poseidon2_gate.rs:312-397 etc.) agree on random
   wires -- the in-circuit one by building the gate-eval circuit, feeding the extension evaluator's values as
   expectations, and requiring witness generation + every constraint of that circuit to hold.
2. The recursive verifier circuit for a small inner circuit: built, its witness generated from a real inner proof,
   proved and verified by the oracle; a tampered inner proof has no witness.
"""
import numpy as np
import pytest

from conftest import P, splitmix_field

GATES = {1: "Constant", 2: "PublicInput", 3: "BaseSum", 4: "U32Interleave", 5: "UninterleaveToU32", 6: "Arithmetic",
         7: "MulExtension", 8: "Exponentiation", 9: "U32Arithmetic", 10: "Poseidon2", 11: "ArithmeticExtension",
         12: "Poseidon", 13: "RandomAccess", 14: "Reducing", 15: "ReducingExtension",
         16: "CosetInterpolation", 17: "PoseidonMds"}


@pytest.mark.parametrize("kind", sorted(GATES))
def test_eval_fns_base_extension_and_circuit_agree(p25, oracle, kind):
    wires = splitmix_field(270, seed=100 + kind).reshape(135, 2)
    consts = splitmix_field(4, seed=200 + kind).reshape(2, 2)
    pih = splitmix_field(4, seed=300 + kind)
    # base vs extension: on wires from the base field the extension evaluator returns the base evaluator's values
    wb = wires.copy()
    wb[:, 1] = 0
    kb = consts.copy()
    kb[:, 1] = 0
    assert (oracle.eval_gate(kind, wb, kb, pih, base=True) == oracle.eval_gate(kind, wb, kb, pih)).all()
    # extension vs circuit
    expect = oracle.eval_gate(kind, wires, consts, pih)
    c = p25.Circuit.build_gate_eval(kind)
    inp = np.concatenate([wires.ravel(), consts.ravel(), pih, expect.ravel()])
    assert int(c.info.num_inputs) == inp.size, GATES[kind]
    oc = oracle.load_circuit(c.to_blob())
    w, st, msg = oc.witness(inp, seed=1)
    assert st == 0, (GATES[kind], msg)
    bad, msg = oc.check_constraints(w)
    assert bad == 0, (GATES[kind], msg)
    # any wrong expectation is caught (first, middle and last constraint)
    for j in {0, expect.shape[0] // 2, expect.shape[0] - 1}:
        wrong = inp.copy()
        k = 270 + 4 + 4 + 2 * j
        wrong[k] = (int(wrong[k]) + 1) % P
        assert oc.witness(wrong, seed=1)[1] == 4, (GATES[kind], j)


def test_gate_eval_rejects_gates_without_evaluator(p25):
    for kind in (18, 99, -1):
        with pytest.raises(p25.P25Error):
            p25.Circuit.build_gate_eval(kind)


@pytest.fixture(scope="module")
def small_recursion(p25, oracle):
    """inner = the `and` gadget circuit (all u32 gates); outer = recursive verifier of one proof of it."""
    inner = p25.Circuit.build_gadget(0, 0)
    oi = oracle.load_circuit(inner.to_blob())
    x, y = 0x0123456789ABCDEF % P, 0x0FEDCBA987654321 % P
    inp = np.array([x, y, (x & y) % P], dtype=np.uint64)
    proof, st, _t, msg = oi.prove(inp, seed=5)
    assert st == 0, msg
    dg, cap = oi.digest()
    assert oi.verify(proof, dg, cap)[0] == 0
    outer = inner.build_recursive_verifier(1, digest=dg, cs_cap=cap)
    return inner, oi, proof, outer


def test_recursive_verifier_of_a_small_circuit(p25, oracle, small_recursion):
    inner, oi, proof, outer = small_recursion
    assert int(outer.info.num_inputs) == int(inner.info.proof_words)
    counts = outer.gate_counts()
    assert any(k.startswith("PoseidonGate") for k in counts) and "ArithmeticExtensionGate { num_ops: 10 }" in counts
    # upstream's verifier gate set (round 3): cap / evaluation selection on RandomAccessGate, reductions with powers of
    # alpha on ReducingGate (base-field terms) and ReducingExtensionGate (extension terms)
    assert any(k.startswith("RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2") for k in counts)
    assert "ReducingGate { num_coeffs: 43 }" in counts and "ReducingExtensionGate { num_coeffs: 32 }" in counts
    oo = oracle.load_circuit(outer.to_blob())
    wires, st, msg = oo.witness(proof, seed=9)
    assert st == 0, msg
    bad, msg = oo.check_constraints(wires)
    assert bad == 0, msg
    outer_proof, st, _t, msg = oo.prove(proof, seed=9)
    assert st == 0, msg
    assert oo.verify(outer_proof)[0] == 0
    print("outer circuit: 2^%d rows, gates %s" % (int(outer.info.degree_bits), counts))


def test_recursive_verifier_rejects_tampered_inner_proofs(oracle, small_recursion):
    inner, oi, proof, outer = small_recursion
    oo = oracle.load_circuit(outer.to_blob())
    n = proof.size
    # a cap word, an opening, a Merkle sibling / leaf word inside the queries, the final polynomial, the PoW witness
    for k in (0, 64 * 3 + 7, 64 * 3 + 2 * (5 + 80 + 135 + 2 + 2 + 18 + 16) + 100, n - 20, n - 1):
        bad = proof.copy()
        bad[k] = (int(bad[k]) + 1) % P
        assert oi.verify(bad)[0] != 0
        assert oo.witness(bad, seed=9)[1] == 4, k


def test_constraints_not_only_generators_reject_a_bad_inner_proof(oracle, small_recursion):
    """The tests above see a corrupted inner proof fail through the witness generator's `connect` conflict.  That a
    CONSTRAINT fails too -- that no gadget of the verifier circuit merely has its generator do the comparing -- is
    shown here: witness generation carries on past the conflict (the partition keeps its first value, so every copy
    constraint holds by construction) and the gate constraints of the resulting witness must then be violated.
    One corruption per check of the recursive verifier: a cap word (Merkle root select), an opening (vanishing
    identity and reduced openings), a leaf word and a sibling (Merkle paths), a FRI evaluation (fold consistency),
    the final polynomial, the PoW witness (range check of the response)."""
    inner, oi, proof, outer = small_recursion
    oo = oracle.load_circuit(outer.to_blob())
    good, st, _m = oo.witness_forced(proof, seed=9)
    assert st == 0 and oo.check_constraints(good)[0] == 0
    n = proof.size
    openings = 64 * 3
    queries = openings + 2 * (5 + 80 + 135 + 2 + 2 + 18 + 16) + 64 * len(range(0))
    spots = {"wires cap": 5, "opening (wires)": openings + 2 * (5 + 80) + 3, "opening (quotient)": openings + 2 * (5 + 80 + 135 + 4 + 18) + 1,
             "query leaf word": queries + 40, "query sibling": queries + 85 + 7, "final poly": n - 4, "pow witness": n - 1}
    for what, k in spots.items():
        bad = proof.copy()
        bad[k] = (int(bad[k]) + 1) % P
        assert oi.verify(bad)[0] != 0, what
        wires, st, _m = oo.witness_forced(bad, seed=9)
        assert st == 4, what
        nbad, msg = oo.check_constraints(wires)
        assert nbad > 0, f"{what}: every gate constraint holds on a witness built from a false inner proof"


def test_recursive_verifier_with_fri_layers(p25, oracle):
    """inner = the plonky3-verifier circuit of a 2^3-row Fibonacci STARK (2^10 rows, all 11 gate types, one FRI
    reduction layer): covers the in-circuit fold (coset interpolation at beta), the layer Merkle proofs and the
    Poseidon2 gate's eval_unfiltered_circuit inside a real verifier."""
    inp, cfg = p25.p3_prove_fibonacci(3, 3, 4)
    inner = p25.Circuit.build_p3_verifier(cfg)
    oi = oracle.load_circuit(inner.to_blob())
    proof, st, _t, msg = oi.prove(inp, seed=2)
    assert st == 0, msg
    dg, cap = oi.digest()
    outer = inner.build_recursive_verifier(1, digest=dg, cs_cap=cap)
    assert any(k.startswith("CosetInterpolationGate { subgroup_bits: 4, degree: 6") for k in outer.gate_counts())
    oo = oracle.load_circuit(outer.to_blob())
    wires, st, msg = oo.witness(proof, seed=1)
    assert st == 0, msg
    bad, msg = oo.check_constraints(wires)
    assert bad == 0, msg
    # a flipped evaluation inside a FRI step of the last query
    tampered = proof.copy()
    tampered[-40] = (int(tampered[-40]) + 1) % P
    assert oo.witness(tampered, seed=1)[1] == 4
    print("outer: 2^%d rows for an inner circuit of 2^%d" % (int(outer.info.degree_bits), int(inner.info.degree_bits)))


def test_recursion_of_recursion(p25, oracle, small_recursion):
    """Depth 2: a circuit verifying a proof of the recursive verifier (whose rows include PoseidonGate and
    ArithmeticExtensionGate, evaluated in-circuit) -- what every inner node of an aggregation tree is."""
    inner, oi, proof, outer = small_recursion
    oo = oracle.load_circuit(outer.to_blob())
    outer_proof, st, _t, msg = oo.prove(proof, seed=9)
    assert st == 0, msg
    dg, cap = oo.digest()
    outer2 = outer.build_recursive_verifier(1, digest=dg, cs_cap=cap)
    o2 = oracle.load_circuit(outer2.to_blob())
    wires, st, msg = o2.witness(outer_proof, seed=3)
    assert st == 0, msg
    bad, msg = o2.check_constraints(wires)
    assert bad == 0, msg
    tampered = outer_proof.copy()
    tampered[200] = (int(tampered[200]) + 1) % P
    assert o2.witness(tampered, seed=3)[1] == 4
    print("depth-2 circuit: 2^%d rows" % int(outer2.info.degree_bits))


def test_recursive_verifier_single_selector_inner(p25, oracle):
    """inner circuit with ONE selector polynomial (four gate types of degree <= 3: no UNUSED_SELECTOR factor in the
    filters) and 2^2 rows -- the smallest shapes the verifier circuit has to handle (no FRI layer, LDE of 32 points,
    Merkle paths of one sibling)."""
    inner = p25.Circuit.build_gadget(8, 0)
    assert int(inner.info.num_selectors) == 1
    oi = oracle.load_circuit(inner.to_blob())
    proof, st, _t, msg = oi.prove(np.array([7, 7, 49], dtype=np.uint64), seed=1)
    assert st == 0, msg
    dg, cap = oi.digest()
    outer = inner.build_recursive_verifier(2, digest=dg, cs_cap=cap)      # two proofs of it at once
    proof2, st, _t, msg = oi.prove(np.array([3, 3, 9], dtype=np.uint64), seed=2)
    assert st == 0, msg
    oo = oracle.load_circuit(outer.to_blob())
    both = np.concatenate([proof, proof2])
    wires, st, msg = oo.witness(both, seed=4)
    assert st == 0, msg
    assert oo.check_constraints(wires)[0] == 0
    bad = both.copy()
    bad[proof.size + 10] = (int(bad[proof.size + 10]) + 1) % P
    assert oo.witness(bad, seed=4)[1] == 4
