"""The circuit-blob boundary with an independent producer: a blob written by tests/blob_writer.py (pure Python,
from the specification in INTEGRATION.md section 5 -- the stand-in for a Rust host that keeps `builder.build::<C>()`,
/root/reference/src/p3/mod.rs:250) is imported with p25_circuit_import and proved with."""
import numpy as np
import pytest

import blob_writer
from conftest import P


def test_python_blob_imports_and_matches_the_products_builder_cpu(p25, oracle):
    """Same circuit built twice -- by the product's C++ builder (gadget 8) and by the independent Python builder:
    identical verifier data (constants/sigmas commitment, circuit digest) and identical proofs from the oracle."""
    blob = blob_writer.connected_inputs_product().to_blob()
    c_py = p25.Circuit.from_blob(blob)
    c_cc = p25.Circuit.build_gadget(8, 0)
    for f in ("degree_bits", "num_rows_used", "num_wires", "num_routed_wires", "num_inputs", "num_generators",
              "num_gate_types", "num_selectors", "num_constants_sigmas", "num_gate_constraints", "proof_words"):
        assert getattr(c_py.info, f) == getattr(c_cc.info, f), f
    assert c_py.gate_counts() == c_cc.gate_counts()
    o_py, o_cc = oracle.load_circuit(blob), oracle.load_circuit(c_cc.to_blob())
    d_py, cap_py = o_py.digest()
    d_cc, cap_cc = o_cc.digest()
    assert (d_py == d_cc).all() and (cap_py == cap_cc).all()
    inp = np.array([6, 6, 36], dtype=np.uint64)
    p_py, st_py, _t, msg = o_py.prove(inp, seed=3)
    p_cc, st_cc, _t, _m = o_cc.prove(inp, seed=3)
    assert st_py == 0 and st_cc == 0, msg
    assert (p_py == p_cc).all()
    assert o_cc.verify(p_py, d_cc, cap_cc)[0] == 0
    # re-export of the imported circuit parses again (import -> export round trip through the C ABI)
    assert p25.Circuit.from_blob(c_py.to_blob()).info.num_generators == c_py.info.num_generators


def test_python_blob_of_a_circuit_the_library_cannot_build_cpu(p25, oracle):
    c = blob_writer.sum_of_products(12)
    blob = c.to_blob()
    circ = p25.Circuit.from_blob(blob)
    assert int(circ.info.num_inputs) == 25 and circ.gate_counts()["ArithmeticGate { num_ops: 20 }"] == 2
    oc = oracle.load_circuit(blob)
    rng = np.random.default_rng(1)
    xs = [int(v) for v in rng.integers(0, P, size=24, dtype=np.uint64)]
    y = sum(xs[2 * i] * xs[2 * i + 1] for i in range(12)) % P
    good = np.array(xs + [y], dtype=np.uint64)
    wires, st, msg = oc.witness(good, seed=1)
    assert st == 0, msg
    assert oc.check_constraints(wires)[0] == 0
    bad = good.copy()
    bad[-1] = (y + 1) % P
    assert oc.witness(bad, seed=1)[1] == 4


@pytest.mark.gpu
def test_python_blob_proves_on_the_gpu(gpu, oracle):
    rng = np.random.default_rng(2)
    for make, n_in in ((blob_writer.connected_inputs_product, 3), (lambda: blob_writer.sum_of_products(12), 25)):
        blob = make().to_blob()
        circ = gpu.Circuit.from_blob(blob)
        oc = oracle.load_circuit(blob)
        if n_in == 3:
            good = np.array([9, 9, 81], dtype=np.uint64)
        else:
            xs = [int(v) for v in rng.integers(0, P, size=24, dtype=np.uint64)]
            good = np.array(xs + [sum(xs[2 * i] * xs[2 * i + 1] for i in range(12)) % P], dtype=np.uint64)
        bad = good.copy()
        bad[-1] = (int(bad[-1]) + 1) % P
        proofs, st = circ.prove(np.stack([good, bad, good]), seeds=[1, 1, 2])
        assert st.tolist() == [0, 4, 0]
        po, sto, _t, msg = oc.prove(good, seed=1)
        assert sto == 0, msg
        assert (proofs[0] == po).all()
        dg, cap = circ.digest()
        do, capo = oc.digest()
        assert (dg == do).all() and (cap == capo).all()
        assert oc.verify(proofs[2], dg, cap)[0] == 0
    # the Python-built twin of gadget 8 has the product builder's digest on the GPU too
    twin = gpu.Circuit.from_blob(blob_writer.connected_inputs_product().to_blob())
    assert (twin.digest()[0] == gpu.Circuit.build_gadget(8, 0).digest()[0]).all()


def reference_gates_inputs(oracle, x, y, z):
    """Inputs of gadget 14 / blob_writer.reference_gates(): the operands and, natively, what the reference's gates compute."""
    spread = lambda v: sum(((v >> i) & 1) << (2 * i) for i in range(32))
    m, xi, yi = x * y % P, spread(x), spread(y)
    ev, od = 0, x                                   # uninterleave(spread(x)): evens (bits 63, 61, ...) 0, odds x
    s = x * y + z
    lo, hi = s & 0xFFFFFFFF, s >> 32
    h = oracle.poseidon2_permute(np.array([m, xi, yi, ev, od, lo, hi, x, 0, 0, 0, 0], dtype=np.uint64))[0]
    return np.array([x, y, z, m, xi, yi, ev, od, lo, hi] + [int(v) for v in h[:4]], dtype=np.uint64)


def test_python_builder_reproduces_a_circuit_with_the_references_four_gates_cpu(p25, oracle):
    """The independent builder extended with U32InterleaveGate, UninterleaveToU32Gate, U32ArithmeticGate and Poseidon2Gate
    (wire layouts, generator payloads and gate ids from the reference's files, the selector GROUPING rule from upstream's
    selectors.rs) builds gadget 14's circuit from the blob specification alone: 8 gate types in two selector groups, 138
    generators -- same verifier data (circuit digest, constants/sigmas cap) and the same proof bytes as the circuit the
    product's C++ builder emits.  Circuit-shape independence for the reference's own gates does not rest on the product's
    builder alone any more ("the oracle receives its circuit from the product's builder")."""
    blob = blob_writer.reference_gates().to_blob()
    c_py, c_cc = p25.Circuit.from_blob(blob), p25.Circuit.build_gadget(14, 0)
    for f in ("degree_bits", "num_rows_used", "num_inputs", "num_generators", "num_gate_types", "num_selectors",
              "num_constants_sigmas", "num_gate_constraints", "proof_words"):
        assert getattr(c_py.info, f) == getattr(c_cc.info, f), f
    assert int(c_py.info.num_selectors) == 2 and int(c_py.info.num_gate_types) == 8
    assert c_py.gate_counts() == c_cc.gate_counts()
    o_py, o_cc = oracle.load_circuit(blob), oracle.load_circuit(c_cc.to_blob())
    (d_py, cap_py), (d_cc, cap_cc) = o_py.digest(), o_cc.digest()
    assert (d_py == d_cc).all() and (cap_py == cap_cc).all()
    rng = np.random.default_rng(14)
    for x, y, z in [(0xFFFFFFFC, 0xFFFFFFFF, 0xFFFFFFFF), (0, 0, 0)] + [tuple(int(v) for v in rng.integers(0, 1 << 32, size=3)) for _ in range(3)]:
        inp = reference_gates_inputs(oracle, x, y, z)
        w, st, msg = o_py.witness(inp, seed=4)
        assert st == 0 and o_py.check_constraints(w)[0] == 0, msg
        p_py, st_py, _t, msg = o_py.prove(inp, seed=4)
        p_cc, st_cc, _t, _m = o_cc.prove(inp, seed=4)
        assert st_py == 0 and st_cc == 0, msg
        assert (p_py == p_cc).all()
        assert o_cc.verify(p_py, d_cc, cap_cc)[0] == 0
        for k in range(3, 14):                       # every expectation is enforced by a copy constraint
            bad = inp.copy()
            bad[k] = (int(bad[k]) + 1) % P
            assert o_py.witness(bad, seed=4)[1] == 4, k


@pytest.mark.gpu
def test_python_built_reference_gates_circuit_proves_on_the_gpu(gpu, oracle):
    blob = blob_writer.reference_gates().to_blob()
    circ, cc = gpu.Circuit.from_blob(blob), gpu.Circuit.build_gadget(14, 0)
    oc = oracle.load_circuit(blob)
    good = reference_gates_inputs(oracle, 0x89ABCDEF, 0x01234567, 0xFFFFFFFF)
    bad = good.copy()
    bad[5] = (int(bad[5]) + 1) % P
    proofs, st = circ.prove(np.stack([good, bad]), seeds=[1, 1])
    assert st.tolist() == [0, 4]
    po, sto, _t, msg = oc.prove(good, seed=1)
    assert sto == 0 and (proofs[0] == po).all(), msg
    assert (circ.digest()[0] == cc.digest()[0]).all()
    p2, st2 = cc.prove(good, seeds=[1])
    assert st2.tolist() == [0] and (p2[0] == proofs[0]).all()
