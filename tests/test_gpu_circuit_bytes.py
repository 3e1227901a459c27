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


def test_independent_reader_on_gpu_written_bytes(gpu, oracle):
    """The symmetric-blind spot closed: `p25_circuit_to_bytes` output parsed by tests/circuit_bytes_reader.py (pure Python,
    from INTEGRATION.md 5a.1), the blob rebuilt from the parse, imported, and compared with the circuit the bytes came
    from -- same circuit digest, same proof bytes -- for gadget circuits that together hold all four of the reference's
    gates (tags 16-19 with their `serialize` payloads) and for a recursive verifier circuit (tags 1, 11, 13, 14, 15)."""
    import circuit_bytes_reader as cr
    from gadget_cases import cases
    by_name = {c[0]: c for c in cases(oracle)}
    seen_tags = set()
    last = None
    for name in ("and", "lsh33", "rev19", "exp19", "ext_arith", "compress"):
        _n, kind, param, vals = by_name[name]
        c = gpu.Circuit.build_gadget(kind, param)
        parsed = cr.parse(c.to_bytes())
        cr.check_reference_payloads(parsed)
        seen_tags |= {t for t, _ in parsed["gates"]}
        twin = gpu.Circuit.from_blob(cr.to_blob(parsed, c.input_target_indices()))
        assert twin.to_blob() == c.to_blob(), name
        dg, cap = c.digest()
        dg2, cap2 = twin.digest()
        assert (dg == dg2).all() and (cap == cap2).all() and parsed["circuit_digest"] == [int(v) for v in dg], name
        assert parsed["cap"] == np.asarray(cap, dtype=np.uint64).reshape(16, 4).tolist()
        inp = np.array(vals, dtype=np.uint64)[None, :]
        p1, s1 = c.prove(inp, seeds=[2])
        p2, s2 = twin.prove(inp, seeds=[2])
        assert s1.tolist() == [0] and s2.tolist() == [0] and (p1 == p2).all(), name
        last = (c, p1)
    assert {16, 17, 18, 19} <= seen_tags, seen_tags          # Poseidon2Gate, U32ArithmeticGate, U32InterleaveGate, UninterleaveToU32Gate
    rc = last[0].build_recursive_verifier(1)
    parsed = cr.parse(rc.to_bytes())
    cr.check_reference_payloads(parsed)
    assert {1, 11, 13, 14, 15} <= {t for t, _ in parsed["gates"]}
    twin = gpu.Circuit.from_blob(cr.to_blob(parsed, rc.input_target_indices()))
    assert twin.to_blob() == rc.to_blob()
    p1, s1 = rc.prove(last[1], seeds=[4])
    p2, s2 = twin.prove(last[1], seeds=[4])
    assert s1.tolist() == [0] and (p1 == p2).all()


def test_fib64_circuit_roundtrip(gpu, fib_circuit, fib_inputs):
    data, _ = roundtrip(gpu, fib_circuit, fib_inputs[None, :], [11])
    assert len(data) > 400 << 20      # leaves of the constants/sigmas tree alone: 2^19 x 85 field elements
