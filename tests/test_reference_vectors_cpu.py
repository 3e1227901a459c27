"""CPU: the remaining vectors the reference's own tests hold for this path (VERDICT r4 "what's missing" 3), on the oracle
and on the circuits the product's host code builds.  GPU twin: tests/test_gpu_reference_vectors.py."""
import numpy as np
import pytest

import reference_vectors as rv
from conftest import P, splitmix_field


def test_interleave_u32_reference_case(p25, oracle):
    """test_interleave_u32 (interleaved_u32.rs:354-382): constant_u32(0xFFFFFFFC) -> interleave_u32 -> public input; the
    proof's public input must be the literal 0x5555...50 and the proof must verify.  param 1 = the test's circuit as
    written (no witness inputs), param 0 = the same value through a witness input."""
    for param, inp in ((1, []), (0, [rv.INTERLEAVE_X])):
        c = p25.Circuit.build_gadget(12, param)
        assert int(c.info.num_inputs) == len(inp) and int(c.info.num_public_inputs) == 1
        assert any(k.startswith("U32InterleaveGate") for k in c.gate_counts())
        oc = oracle.load_circuit(c.to_blob())
        proof, st, _t, msg = oc.prove(np.array(inp, dtype=np.uint64), seed=1)
        assert st == 0, msg
        assert [int(v) for v in c.public_inputs(proof)] == [rv.INTERLEAVE_EXPECTED]
        assert oc.verify(proof)[0] == 0
        bad = proof.copy()
        bad[-1] = np.uint64(rv.INTERLEAVE_EXPECTED ^ 1)
        assert oc.verify(bad)[0] != 0


def test_uninterleave_to_u32_reference_case(p25, oracle):
    """test_uninterleave_to_u32 (interleaved_u32.rs:388-417): evens 0xC0000000, odds 0xFFFFFFFF."""
    for param, inp in ((1, []), (0, [rv.UNINTERLEAVE_X])):
        c = p25.Circuit.build_gadget(13, param)
        assert int(c.info.num_inputs) == len(inp) and int(c.info.num_public_inputs) == 2
        assert any(k.startswith("UninterleaveToU32Gate") for k in c.gate_counts())
        oc = oracle.load_circuit(c.to_blob())
        proof, st, _t, msg = oc.prove(np.array(inp, dtype=np.uint64), seed=1)
        assert st == 0, msg
        assert [int(v) for v in c.public_inputs(proof)] == [rv.UNINTERLEAVE_EVENS_EXPECTED, rv.UNINTERLEAVE_ODDS_EXPECTED]
        assert oc.verify(proof)[0] == 0


@pytest.mark.parametrize("x", [0, 1, 0x01234567, 0x89ABCDEF, 0xFFFFFFFF, 0xAAAAAAAA])
def test_interleave_then_uninterleave_round_trip(p25, oracle, x):
    """Native expectations as the reference's gadget tests compute them: interleave spreads bit i to bit 2i."""
    ci, cu = p25.Circuit.build_gadget(12, 0), p25.Circuit.build_gadget(13, 0)
    oi, ou = oracle.load_circuit(ci.to_blob()), oracle.load_circuit(cu.to_blob())
    spread = sum(((x >> i) & 1) << (2 * i) for i in range(32))
    pr, st, _t, msg = oi.prove(np.array([x], dtype=np.uint64), seed=2)
    assert st == 0 and int(ci.public_inputs(pr)[0]) == spread, msg
    # uninterleave(spread(x) + 2 spread(y)): evens = bits 63, 61, ... = y, odds = x
    y = (x * 2654435761 + 12345) & 0xFFFFFFFF
    both = spread + 2 * sum(((y >> i) & 1) << (2 * i) for i in range(32))
    if both >= P:
        return
    pr, st, _t, msg = ou.prove(np.array([both], dtype=np.uint64), seed=2)
    assert st == 0, msg
    assert [int(v) for v in cu.public_inputs(pr)] == [y, x]


def _row(wires114, filler_seed=5):
    """A 135-wire row in the extension-evaluator layout [135][2]: the gate's wires from the base field, the unused ones arbitrary."""
    w = np.zeros((135, 2), dtype=np.uint64)
    w[:, 0] = splitmix_field(135, seed=filler_seed)
    w[:rv.U32_ARITH_WIRES, 0] = wires114
    return w


def test_u32_arithmetic_gate_constraint_and_canonicity(oracle):
    """test_gate_constraint / test_canonicity (arithmetic_u32.rs:554-627) on the oracle's evaluator
    (oracle/ref_gates.h, both the base-field and the extension form): `get_wires` values satisfy every constraint; the
    non-canonical addend 0xFFFFFFFF00000001 (output_high = u32::MAX, output_low = 1) must violate one."""
    consts = np.zeros((2, 2), dtype=np.uint64)
    pih = splitmix_field(4, seed=9)                     # the test passes HashOut::rand(): the gate ignores it
    for name, wires, satisfied in rv.u32_arithmetic_cases():
        w = _row(wires)
        for base in (True, False):
            out = oracle.eval_gate(9, w, consts, pih, base=base)
            assert out.shape[0] == 3 * (2 + 32 + 2)           # num_constraints = num_ops * (4 + num_limbs) (arithmetic_u32.rs:167-169)
            assert (not out.any()) == satisfied, (name, base, np.argwhere(out)[:4])
        if not satisfied:
            # which one: hi_not_max * lo of every op = (0 * 0 - 1) * 1 = -1
            out = oracle.eval_gate(9, w, consts, pih, base=True)
            assert [int(out[36 * i, 0]) for i in range(3)] == [P - 1] * 3
            assert not np.delete(out, [0, 36, 72], axis=0).any()


def test_u32_arithmetic_in_circuit_evaluator_on_the_reference_wires(p25, oracle):
    """The same wires through `eval_unfiltered_circuit` (arithmetic_u32.rs:178-245): the gate-eval circuit accepts the
    expectation "all constraints zero" for get_wires values and has NO witness for it on the canonicity wires."""
    c = p25.Circuit.build_gate_eval(9)
    oc = oracle.load_circuit(c.to_blob())
    consts, pih = np.zeros((2, 2), dtype=np.uint64), splitmix_field(4, seed=9)
    for name, wires, satisfied in rv.u32_arithmetic_cases(n_random=2):
        w = _row(wires)
        zeros = np.zeros((108, 2), dtype=np.uint64)
        inp = np.concatenate([w.ravel(), consts.ravel(), pih, zeros.ravel()])
        _w, st, msg = oc.witness(inp, seed=1)
        assert (st == 0) == satisfied, (name, st, msg)
        if not satisfied:
            expect = oracle.eval_gate(9, w, consts, pih)
            ok_inp = np.concatenate([w.ravel(), consts.ravel(), pih, expect.ravel()])
            wt, st, msg = oc.witness(ok_inp, seed=1)
            assert st == 0 and oc.check_constraints(wt)[0] == 0, msg


def test_poseidon2_gate_wire_indices(p25, oracle):
    """`wire_indices` (poseidon2_gate.rs:553-565) pinned behaviourally on both implementations:
    inputs 0..11 / outputs 12..23 on a satisfied row of the compress gadget (product builder + oracle witness generator:
    outputs = the artifact-pinned Poseidon2 permutation of the inputs), WIRE_SWAP 24 and wire_delta 25..28 through the
    first five constraints of the evaluator (poseidon2_gate.rs:157-171) on arbitrary wires."""
    pins = rv.POSEIDON2_WIRE_PINS
    c = p25.Circuit.build_gadget(5, 0)
    oc = oracle.load_circuit(c.to_blob())
    l, r = splitmix_field(4, seed=31), splitmix_field(4, seed=32)
    state = np.concatenate([l, r, np.zeros(4, dtype=np.uint64)])
    perm = oracle.poseidon2_permute(state)[0]
    wires, st, msg = oc.witness(np.concatenate([l, r, perm[:4]]), seed=1)
    assert st == 0, msg
    rows = [row for row in range(wires.shape[1]) if (wires[pins["wire_input(0)"]:pins["wire_input(11)"] + 1, row] == state).all()
            and wires[:12, row].any()]
    assert len(rows) == 1
    row = rows[0]
    assert (wires[pins["wire_output(0)"]:pins["wire_output(11)"] + 1, row] == perm).all()
    assert int(wires[pins["WIRE_SWAP"], row]) == 0
    assert not wires[pins["wire_delta(0)"]:pins["wire_delta(3)"] + 1, row].any()
    # the evaluator: constraint 0 = swap (swap - 1), constraints 1..4 = swap (in[i+4] - in[i]) - delta_i
    w = splitmix_field(270, seed=77).reshape(135, 2)
    w[:, 1] = 0
    out = oracle.eval_gate(10, w, np.zeros((2, 2), dtype=np.uint64), np.zeros(4, dtype=np.uint64), base=True)
    swap = int(w[pins["WIRE_SWAP"], 0])
    assert int(out[0, 0]) == swap * (swap - 1) % P
    for i in range(4):
        delta = int(w[pins["wire_delta(0)"] + i, 0])
        assert int(out[1 + i, 0]) == (swap * (int(w[i + 4, 0]) - int(w[i, 0])) - delta) % P
    assert pins["wire_delta(0)"] + 3 == pins["wire_delta(3)"]
    # with swap = 1 and consistent deltas the permutation runs on the SWAPPED inputs: outputs = perm(r || l || cap)
    w2 = np.zeros((135, 2), dtype=np.uint64)
    w2[:12, 0] = state
    w2[pins["WIRE_SWAP"], 0] = 1
    for i in range(4):
        w2[pins["wire_delta(0)"] + i, 0] = (int(r[i]) - int(l[i])) % P
    out = oracle.eval_gate(10, w2, np.zeros((2, 2), dtype=np.uint64), np.zeros(4, dtype=np.uint64), base=True)
    assert not out[:5].any()
