"""CPU: the host-side circuit builder (product) against the reference's artifact, checked by the oracle.

The strongest pin the reference tree offers for the whole in-circuit plonky3 verifier: witness
generation on artifacts/proof_fibonacci.json succeeds only if every `connect` in src/p3 holds
(Merkle roots == commitments commit.rs:125-127, folded evaluations == final_poly verifier.rs:413,
constraint/quotient identity verifier.rs:239, PoW challenger.rs:159-168)."""
import json

import numpy as np
import pytest

from conftest import ARTIFACT, P
import p3json


def test_circuit_shape(fib_circuit):
    info = fib_circuit.info
    assert int(info.degree_bits) == 16           # n = 2^16
    assert int(info.num_inputs) == 15751         # src/p3/serde/proof.rs:357-383
    assert int(info.num_wires) == 135 and int(info.num_routed_wires) == 80
    assert int(info.num_selectors) == 3 and int(info.num_constants_sigmas) == 85
    assert int(info.num_gate_constraints) == 134  # UninterleaveToU32Gate, uninterleave_to_u32.rs:268-270
    assert int(info.proof_words) == 19861
    counts = fib_circuit.gate_counts()
    p2 = [v for k, v in counts.items() if k.startswith("Poseidon2Gate")]
    assert p2 == [4317]                           # 4,317 Poseidon2 permutations per proof
    order = list(counts)
    assert order[0] == "NoopGate" and order[-1].startswith("Poseidon2Gate")
    assert order.index("ConstantGate { num_consts: 2 }") < order.index("PublicInputGate")


def test_json_loader_matches_python_flatten(p25, fib_inputs):
    text = open(ARTIFACT).read()
    inp, cfg = p25.p3_proof_from_json(text)
    assert (inp == fib_inputs).all()
    assert (cfg.log_trace_height, cfg.trace_width, cfg.opening_matrix_log_max_height, cfg.quotient_opened_len,
            cfg.degree_bits, cfg.num_queries, cfg.log_quotient_degree) == (6, 3, 7, 2, 6, 100, 0)
    with pytest.raises(p25.P25Error) as e:
        p25.p3_proof_from_json(text[:-20])
    assert e.value.status == 8
    bad = text.replace('"value":', '"value": 18446744069414584321, "x":', 1)  # non-canonical element
    with pytest.raises(p25.P25Error):
        p25.p3_proof_from_json(bad)


def test_blob_roundtrip(p25, fib_circuit, fib_blob):
    c2 = p25.Circuit.from_blob(fib_blob)
    assert c2.to_blob() == fib_blob
    with pytest.raises(p25.P25Error):
        p25.Circuit.from_blob(fib_blob[:1000])


def test_witness_on_artifact_satisfies_circuit(fib_oracle, fib_inputs):
    wires, st, msg = fib_oracle.witness(fib_inputs, seed=7)
    assert st == 0, msg
    bad, msg = fib_oracle.check_constraints(wires)
    assert bad == 0, msg
    # golden transcript values of the inner proof appear on Poseidon2 output wires
    alpha0 = 13582184458757534322
    assert (wires[12:24] == np.uint64(alpha0)).any()


def test_bad_inner_proof_is_rejected(fib_oracle, fib_inputs):
    # flipping any element of the plonky3 proof must trip a copy constraint (upstream panics)
    for pos in (0, 40, 9000, 15750):
        inp = fib_inputs.copy()
        inp[pos] = (int(inp[pos]) + 1) % P
        _w, st, msg = fib_oracle.witness(inp, seed=0)
        assert st == 4 and "set twice" in msg
