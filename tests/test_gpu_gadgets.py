"""GPU parity on the small gadget circuits: witness, proof bytes and digest equal the oracle's."""
import numpy as np
import pytest

from gadget_cases import cases, P

pytestmark = pytest.mark.gpu


def test_gadgets_gpu_equals_oracle(gpu, oracle):
    for name, kind, param, vals in cases(oracle):
        c = gpu.Circuit.build_gadget(kind, param)
        oc = oracle.load_circuit(c.to_blob())
        inp = np.array(vals, dtype=np.uint64)
        wg, st = c.witness(inp, seed=11)
        wo, sto, msg = oc.witness(inp, seed=11)
        assert st == 0 and sto == 0, (name, msg)
        assert (wg == wo).all(), name
        dg, capg = c.digest()
        do, capo = oc.digest()
        assert (dg == do).all() and (capg == capo).all(), name
        wrong = inp.copy()
        wrong[-1] = (int(wrong[-1]) + 1) % P
        proofs, sts = c.prove(np.stack([inp, wrong, inp]), seeds=[11, 11, 12])
        assert sts.tolist() == [0, 4, 0], name
        po, sto, _tm, msg = oc.prove(inp, seed=11)
        assert sto == 0, (name, msg)
        diff = np.nonzero(proofs[0] != po)[0]
        assert diff.size == 0, (name, diff[:5])
        assert oc.verify(proofs[2], dg, capg)[0] == 0, name


def test_two_connected_inputs_with_different_values_conflict(gpu, oracle):
    """inputs sharing a copy-constraint partition are compared, not raced (upstream panics
    "Partition ... was set twice with different values")."""
    c = gpu.Circuit.build_gadget(8, 0)
    oc = oracle.load_circuit(c.to_blob())
    good = np.array([5, 5, 25], dtype=np.uint64)
    bad = np.array([5, 6, 30], dtype=np.uint64)      # a * b is right, but a != b
    bad2 = np.array([6, 5, 30], dtype=np.uint64)
    proofs, st = c.prove(np.stack([good, bad, bad2, good]), seeds=[1, 1, 1, 1])
    assert st.tolist() == [0, 4, 4, 0]
    assert (proofs[0] == proofs[3]).all()
    assert oc.witness(bad, seed=1)[1] == 4 and oc.witness(good, seed=1)[1] == 0
