"""GPU twin of tests/test_reference_vectors_cpu.py: the reference's own test vectors through the HIP path (C ABI), byte-equal
to the oracle's proofs and equal to the reference's literal expectations."""
import numpy as np
import pytest

import reference_vectors as rv
from conftest import P, splitmix_field

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,param,inp,expected", [
    (12, 1, [], [rv.INTERLEAVE_EXPECTED]),
    (12, 0, [rv.INTERLEAVE_X], [rv.INTERLEAVE_EXPECTED]),
    (13, 1, [], [rv.UNINTERLEAVE_EVENS_EXPECTED, rv.UNINTERLEAVE_ODDS_EXPECTED]),
    (13, 0, [rv.UNINTERLEAVE_X], [rv.UNINTERLEAVE_EVENS_EXPECTED, rv.UNINTERLEAVE_ODDS_EXPECTED]),
])
def test_interleave_and_uninterleave_reference_cases_on_the_gpu(gpu, oracle, kind, param, inp, expected):
    """test_interleave_u32 / test_uninterleave_to_u32 (interleaved_u32.rs:354-417): witness generation, commitments,
    quotient and FRI of the test's circuit on the GPU; the proof's public inputs are the reference's literals, the bytes
    the oracle's, and the oracle's verifier accepts.  (The circuits have 2^2 rows: the smallest shape the prover sees.)"""
    c = gpu.Circuit.build_gadget(kind, param)
    oc = oracle.load_circuit(c.to_blob())
    x = np.array(inp, dtype=np.uint64).reshape(1, -1)
    proofs, st = c.prove(np.concatenate([x, x]), seeds=[1, 2])
    assert st.tolist() == [0, 0]
    dg, cap = c.digest()
    for i, seed in enumerate((1, 2)):
        po, sto, _t, msg = oc.prove(x[0], seed=seed)
        assert sto == 0, msg
        assert (proofs[i] == po).all(), np.nonzero(proofs[i] != po)[0][:8]
        assert [int(v) for v in c.public_inputs(proofs[i])] == expected
        assert oc.verify(proofs[i], dg, cap)[0] == 0


@pytest.fixture(scope="module")
def small(gpu, oracle):
    """plonky3-verifier circuit of a 2^3-row Fibonacci STARK: 2^10 rows holding every gate type of the fib-64 circuit."""
    import circuit_bytes_reader as cbr
    inp, cfg = gpu.p3_prove_fibonacci(3, 3, 4)
    c = gpu.Circuit.build_p3_verifier(cfg)
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=3)
    assert st == 0, msg
    parsed = cbr.parse(c.to_bytes())
    u32_rows = sorted({pl[1] for tag, pl in parsed["generators"] if tag == 25})    # U32ArithmeticGenerator (num_ops, row, i)
    assert u32_rows
    return c, oc, wires, u32_rows


def test_u32_arithmetic_reference_wires_through_the_quotient_kernel(small):
    """test_gate_constraint / test_canonicity (arithmetic_u32.rs:554-627) on `k_quotient`: the reference's `get_wires`
    values written into a U32ArithmeticGate row of a real witness.  The GPU's quotient chunks equal the oracle's in every
    case; the oracle's row-by-row check says the gate's constraints hold for the satisfied cases (what it then reports
    is the copy constraint the foreign values break) and that constraint 0 of that row is p - 1 for the non-canonical
    addend -- and the GPU's quotient differs from the satisfied one in that case, i.e. the kernel sees the violation."""
    c, oc, wires, u32_rows = small
    betas, gammas, alphas = splitmix_field(6, seed=41).reshape(3, 2)
    row = u32_rows[len(u32_rows) // 2]
    assert oc.check_constraints(wires)[0] == 0
    zs = oc.partial_products(wires, betas, gammas)
    quotients = {}
    for name, vals, satisfied in rv.u32_arithmetic_cases(n_random=2):
        w = wires.copy()
        w[:rv.U32_ARITH_WIRES, row] = vals
        bad, msg = oc.check_constraints(w)
        if satisfied:
            assert "gate kind" not in msg, (name, msg)
        else:
            assert msg.startswith(f"row {row} gate kind 9 constraint 0 = {P - 1}"), msg
        qg = c.quotient(w, zs, betas, gammas, alphas)
        qo = oc.quotient(w, zs, betas, gammas, alphas)
        assert (qg == qo).all(), (name, np.argwhere(qg != qo)[:4])
        quotients[name] = qg
    # the non-canonical case differs from the canonical wires of the same statement (0 * 0 + 0: output 0, limbs 0, inverse of u32::MAX)
    w = wires.copy()
    w[:rv.U32_ARITH_WIRES, row] = rv.u32_arithmetic_get_wires([0] * 3, [0] * 3, [0] * 3)
    assert "gate kind" not in oc.check_constraints(w)[1]
    q0 = c.quotient(w, zs, betas, gammas, alphas)
    assert (q0 == oc.quotient(w, zs, betas, gammas, alphas)).all() and (q0 != quotients["canonicity"]).any()


def test_u32_arithmetic_reference_wires_through_the_in_circuit_evaluator_on_the_gpu(gpu, oracle):
    """`eval_unfiltered_circuit` of U32ArithmeticGate (arithmetic_u32.rs:178-245) as a circuit proved on the GPU: "all
    constraints are zero" is provable for get_wires values (bytes = the oracle's) and has no witness for test_canonicity's."""
    c = gpu.Circuit.build_gate_eval(9)
    oc = oracle.load_circuit(c.to_blob())
    consts, pih = np.zeros((2, 2), dtype=np.uint64), splitmix_field(4, seed=9)
    rows = []
    cases = rv.u32_arithmetic_cases(n_random=2)
    for name, wires, satisfied in cases:
        w = np.zeros((135, 2), dtype=np.uint64)
        w[:, 0] = splitmix_field(135, seed=5)
        w[:rv.U32_ARITH_WIRES, 0] = wires
        rows.append(np.concatenate([w.ravel(), consts.ravel(), pih, np.zeros(216, dtype=np.uint64)]))
    proofs, st = c.prove(np.stack(rows), seeds=list(range(len(rows))))
    assert st.tolist() == [0 if sat else 4 for _n, _w, sat in cases]
    for i, (name, _w, sat) in enumerate(cases):
        if sat:
            po, sto, _t, msg = oc.prove(rows[i], seed=i)
            assert sto == 0 and (proofs[i] == po).all(), (name, msg)


def test_poseidon2_gate_wire_indices_on_the_gpu(gpu, oracle):
    """`wire_indices` (poseidon2_gate.rs:553-565) on the device witness generator and the quotient kernel's Poseidon2
    evaluator: the GPU's witness of the compress gadget holds inputs at 0..11, the permutation's outputs at 12..23, swap
    at 24, deltas at 25..28 (= the oracle's witness, itself pinned in the CPU twin), and perturbing exactly those wires of
    that row changes the quotient the same way on both sides."""
    pins = rv.POSEIDON2_WIRE_PINS
    c = gpu.Circuit.build_gadget(5, 0)
    oc = oracle.load_circuit(c.to_blob())
    l, r = splitmix_field(4, seed=31), splitmix_field(4, seed=32)
    state = np.concatenate([l, r, np.zeros(4, dtype=np.uint64)])
    perm = oracle.poseidon2_permute(state)[0]
    inp = np.concatenate([l, r, perm[:4]])
    wg, st = c.witness(inp, seed=1)
    wo, sto, msg = oc.witness(inp, seed=1)
    assert st == 0 and sto == 0 and (wg == wo).all(), msg
    rows = [row for row in range(wg.shape[1]) if (wg[:12, row] == state).all() and wg[:12, row].any()]
    assert len(rows) == 1
    row = rows[0]
    assert (wg[pins["wire_output(0)"]:pins["wire_output(11)"] + 1, row] == perm).all()
    assert int(wg[pins["WIRE_SWAP"], row]) == 0 and not wg[pins["wire_delta(0)"]:pins["wire_delta(3)"] + 1, row].any()
    betas, gammas, alphas = splitmix_field(6, seed=43).reshape(3, 2)
    zs = oc.partial_products(wg, betas, gammas)
    base = c.quotient(wg, zs, betas, gammas, alphas)
    assert (base == oc.quotient(wg, zs, betas, gammas, alphas)).all()
    for col in sorted(set(pins.values())):
        w = wg.copy()
        w[col, row] = (int(w[col, row]) + 1) % P
        qg, qo = c.quotient(w, zs, betas, gammas, alphas), oc.quotient(w, zs, betas, gammas, alphas)
        assert (qg == qo).all() and (qg != base).any(), col
