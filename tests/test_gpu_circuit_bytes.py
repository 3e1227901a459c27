"""GPU: upstream's binary circuit form: CircuitData::to_bytes -> from_bytes gives the same circuit --
same blob, same circuit digest (stored and recomputed), same proof bytes -- for a gadget circuit, a recursive verifier
circuit (PoseidonGate / ArithmeticExtensionGate) and the fib-64 plonky3-verifier circuit itself (~600 MB of bytes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def roundtrip(gpu, c, inputs, seeds):
    data = c.to_bytes()
    c2, stored = gpu.Circuit.from_bytes(data, c.input_target_indices())
    assert c2.to_blob() == c.to_blob(), "from_bytes(to_bytes(circuit)) is not the circuit"
    dg, cap = c.digest()
    dg2, cap2 = c2.digest()
    assert (stored == dg).all() and (dg2 == dg).all() and (cap2 == cap).all()
    p1, s1 = c.prove(inputs, seeds=seeds)
    p2, s2 = c2.prove(inputs, seeds=seeds)
    assert s1.tolist() == [0] * len(seeds) and s2.tolist() == s1.tolist() and (p1 == p2).all()
    return data, p1


def test_gadget_and_recursive_circuits_roundtrip(gpu, oracle):
    from gadget_cases import cases
    name, kind, param, vals = [c for c in cases(oracle) if c[0] == "compress"][0]
    c = gpu.Circuit.build_gadget(kind, param)
    inp = np.array(vals, dtype=np.uint64)[None, :]
    data, proofs = roundtrip(gpu, c, inp, [3])
    # malformed input is refused, never crashes
    for cut in (0, 7, len(data) // 3, len(data) - 1):
        with pytest.raises(gpu.P25Error):
            gpu.Circuit.from_bytes(data[:cut], c.input_target_indices())
    with pytest.raises(gpu.P25Error):
        gpu.Circuit.from_bytes(data + b"\0", c.input_target_indices())
    # a verifier circuit of that gadget's proofs: the recursion gate set goes through the same serializer
    rc = c.build_recursive_verifier(1)
    roundtrip(gpu, rc, proofs, [5])


def test_fib64_circuit_roundtrip(gpu, fib_circuit, fib_inputs):
    data, _ = roundtrip(gpu, fib_circuit, fib_inputs[None, :], [11])
    assert len(data) > 400 << 20      # leaves of the constants/sigmas tree alone: 2^19 x 85 field elements
