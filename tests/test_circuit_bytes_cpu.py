"""CPU: the reader of upstream's binary circuit form (`CircuitData::from_bytes`, restated in circuit_bytes.cpp) on a
fixture written by this library's `to_bytes` on the GPU box (tools/gen_circuit_bytes_golden.py; the writer needs the
device for the constants/sigmas commitment): the parsed circuit is the one the builder produces, the serializer tags
and payloads are where the format says, and damaged input is refused with a status, never a crash."""
import os
import struct

import numpy as np
import pytest

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "circuit_data_gadget_and.bin")


def test_from_bytes_rebuilds_the_builders_circuit(p25):
    data = open(GOLDEN, "rb").read()
    built = p25.Circuit.build_gadget(0, 0)                      # and(x, y), src/p3/mod.rs:271-310 in miniature
    c, stored_digest = p25.Circuit.from_bytes(data, built.input_target_indices())
    assert c.to_blob() == built.to_blob()
    assert int(c.info.degree_bits) == 4 and int(c.info.num_generators) == int(built.info.num_generators)
    assert (stored_digest < np.uint64(p25.P)).all() and stored_digest.any()
    # the verifier data at the tail: cap height, 16 cap hashes, the digest again
    tail = data[-(8 + 16 * 32 + 32):]
    assert struct.unpack_from("<Q", tail, 0)[0] == 4
    assert np.frombuffer(tail[-32:], dtype=np.uint64).tolist() == stored_digest.tolist()
    # CommonCircuitData starts with the config: num_wires, num_routed_wires, num_constants, security_bits, ...
    assert struct.unpack_from("<4Q", data, 0) == (135, 80, 2, 100)


def test_independent_reader_rebuilds_the_blob_from_the_bytes(p25):
    """tests/circuit_bytes_reader.py (pure Python, written from INTEGRATION.md 5a.1) parses the same bytes: every byte is
    consumed, the gate list carries the serializer tags and payloads the reference's `serialize` bodies write
    (interleave_u32.rs:237-247 -> tag 18, num_ops 3; uninterleave_to_u32.rs:272-283 -> tag 19, num_ops 2), their generators
    `num_ops, row, i` (interleave_u32.rs:340-360, uninterleave_to_u32.rs:396-412) on rows of their gates, and the circuit
    blob REBUILT from the parse imports to the builder's circuit -- the writer is no longer checked by its own reader only."""
    import circuit_bytes_reader as cr
    data = open(GOLDEN, "rb").read()
    c = cr.parse(data)
    rows = cr.check_reference_payloads(c)
    assert rows["U32InterleaveGate"] == 2 and rows["UninterleaveToU32Gate"] == 1 and rows["PublicInputGate"] == 1
    assert [t for t, _ in c["gates"]] == [9, 3, 12, 2, 18, 19, 0]          # sorted by (degree, id), upstream's tag numbers
    assert dict(c["gates"])[18] == (3,) and dict(c["gates"])[19] == (2,)
    gens = {}
    for tag, pl in c["generators"]:
        gens.setdefault(tag, []).append(pl)
    assert len(gens[19]) == 131                                            # RandomValueGenerators of the PublicInputGate row
    assert sorted(pl[2] for pl in gens[26]) == [0, 0, 1, 2] and all(pl[0] == 3 for pl in gens[26])
    assert sorted(pl[2] for pl in gens[27]) == [0, 1] and all(pl[0] == 2 for pl in gens[27])
    assert c["circuit_digest"] == c["verifier_digest"] and c["cap"] == c["verifier_cap"]
    built = p25.Circuit.build_gadget(0, 0)
    blob = cr.to_blob(c, built.input_target_indices())
    twin = p25.Circuit.from_blob(blob)
    assert twin.to_blob() == built.to_blob()
    # the reader notices what the format says must hold: a swapped pair of fields in the config is not a circuit
    bad = bytearray(data)
    bad[0:8], bad[8:16] = data[8:16], data[0:8]                            # num_wires <-> num_routed_wires
    with pytest.raises(AssertionError):
        cr.to_blob(cr.parse(bytes(bad)), built.input_target_indices())


def test_from_bytes_survives_corruption(p25):
    data = open(GOLDEN, "rb").read()
    built = p25.Circuit.build_gadget(0, 0)
    targets = built.input_target_indices()
    rng = np.random.default_rng(5)
    rejected = 0
    for trial in range(120):
        b = bytearray(data)
        kind = trial % 3
        if kind == 0:
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:   # damage the structured front (config, gate list, generators), where every byte means something
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, 6000))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(0, len(b) - 64))
            del b[i:i + int(rng.integers(1, 64))]
        try:
            c, _dg = p25.Circuit.from_bytes(bytes(b), targets)
            c.close()
        except p25.P25Error as e:
            assert e.status == 1
            rejected += 1
    assert rejected > 90
    with pytest.raises(p25.P25Error):
        p25.Circuit.from_bytes(data, np.array([1 << 30], dtype=np.uint32))   # input target out of range
