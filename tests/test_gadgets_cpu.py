"""CPU: small gadget circuits (reference gadget tests, src/p3/mod.rs:271-494, commit.rs:173-198)
built by the product's host code and proven + verified by the oracle.  Exercises the prover at
degree_bits 4..7: zero FRI layers, single-pass NTTs, caps as tall as the trees."""
import numpy as np
import pytest

from gadget_cases import cases, P


def test_gadgets_oracle_prove_verify(p25, oracle):
    seen_bits = set()
    for name, kind, param, vals in cases(oracle):
        c = p25.Circuit.build_gadget(kind, param)
        assert int(c.info.num_inputs) == len(vals), name
        oc = oracle.load_circuit(c.to_blob())
        inp = np.array(vals, dtype=np.uint64)
        wires, st, msg = oc.witness(inp, seed=3)
        assert st == 0, (name, msg)
        bad, msg = oc.check_constraints(wires)
        assert bad == 0, (name, msg)
        proof, st, _tm, msg = oc.prove(inp, seed=3)
        assert st == 0, (name, msg)
        code, msg = oc.verify(proof)
        assert code == 0, (name, msg)
        seen_bits.add(oc.degree_bits)
        # wrong expectation -> the failing connect (upstream panics)
        wrong = inp.copy()
        wrong[-1] = (int(wrong[-1]) + 1) % P
        _w, st, msg = oc.witness(wrong, seed=3)
        assert st == 4, name
        # a tampered proof is rejected
        bad_proof = proof.copy()
        bad_proof[len(bad_proof) // 2] ^= np.uint64(1)
        assert oc.verify(bad_proof)[0] != 0, name
    assert len(seen_bits) >= 2


def test_gadget_bad_params(p25):
    for kind, param in ((2, 64), (3, 32), (4, 0), (10, 0), (10, 201), (99, 0)):
        with pytest.raises(p25.P25Error):
            p25.Circuit.build_gadget(kind, param)
