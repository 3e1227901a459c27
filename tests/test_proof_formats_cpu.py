"""CPU: the proof's wire formats either side of the path: upstream's binary
`ProofWithPublicInputs::to_bytes()` (restated; util/serialization.rs) and the serde JSON of src/p3/mod.rs:261 agree with
each other and round-trip, on a proof produced by the oracle."""
import json
import struct

import numpy as np
import pytest

from conftest import P


@pytest.fixture(scope="module")
def small_proof(p25, oracle):
    inp, cfg = p25.p3_prove_fibonacci(3, 3, 4)
    c = p25.Circuit.build_p3_verifier(cfg)          # 2^10 rows: one FRI layer, 28 query rounds
    oc = oracle.load_circuit(c.to_blob())
    proof, st, _t, msg = oc.prove(inp, seed=1)
    assert st == 0, msg
    return c, proof


def test_bytes_round_trip_and_layout(small_proof):
    c, proof = small_proof
    data = c.proof_to_bytes(proof)
    js = json.loads(c.proof_to_json(proof))
    n_layers = len(js["proof"]["opening_proof"]["commit_phase_merkle_caps"])
    n_paths = 28 * (4 + n_layers)                  # 4 initial oracles + the FRI layers, per query
    assert n_layers == 2 and len(data) == 8 * proof.size + n_paths   # one length byte per Merkle proof
    assert (c.proof_from_bytes(data) == proof).all()
    # first words are the wires cap, little-endian
    assert struct.unpack_from("<4Q", data, 0) == tuple(int(x) for x in proof[:4])
    # the first Merkle proof: after 3 caps, the openings, 1 FRI cap and the first leaf row (constants/sigmas)
    n_open = sum(len(js["proof"]["openings"][k]) for k in ("constants", "plonk_sigmas", "wires", "plonk_zs",
                                                           "plonk_zs_next", "partial_products", "quotient_polys"))
    off = 8 * (3 * 64 + 2 * n_open + n_layers * 64 + len(js["proof"]["opening_proof"]["query_round_proofs"][0]
                                              ["initial_trees_proof"]["evals_proofs"][0][0]))
    sib = js["proof"]["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"]["evals_proofs"][0][1]["siblings"]
    assert data[off] == len(sib) == 10 + 3 - 4
    assert struct.unpack_from("<4Q", data, off + 1) == tuple(sib[0]["elements"])
    assert struct.unpack_from("<Q", data, len(data) - 8)[0] == js["proof"]["opening_proof"]["pow_witness"]


def test_from_bytes_rejects_malformed(p25, small_proof):
    c, proof = small_proof
    data = bytearray(c.proof_to_bytes(proof))
    for mutate in (lambda d: d[:-1], lambda d: d + b"\0", lambda d: d[:8 * 300] + bytes([d[8 * 300] ^ 0xFF]) + d[8 * 300 + 1:]):
        bad = bytes(mutate(bytes(data)))
        if bad == bytes(data):
            continue
        try:
            out = c.proof_from_bytes(bad)
        except p25.P25Error:
            continue
        assert (out != proof).any()                # a flipped low byte of a field element still parses: different proof
    noncanon = bytearray(data)
    noncanon[0:8] = struct.pack("<Q", P)
    with pytest.raises(p25.P25Error):
        c.proof_from_bytes(bytes(noncanon))
    path_len = bytearray(data)
    js = json.loads(c.proof_to_json(proof))
    n_open = sum(len(v) for v in js["proof"]["openings"].values())
    leaf0 = len(js["proof"]["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"]["evals_proofs"][0][0])
    off = 8 * (3 * 64 + 2 * n_open + 2 * 64 + leaf0)
    path_len[off] ^= 1
    with pytest.raises(p25.P25Error):
        c.proof_from_bytes(bytes(path_len))
