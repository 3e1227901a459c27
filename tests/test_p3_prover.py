"""CPU: the native plonky3 Fibonacci prover (plonky2.5_amd/csrc/p3_prover.cpp).

Golden pin: with the artifact's parameters it reproduces every one of the 15,751 field elements of
the reference's artifacts/proof_fibonacci.json, including its proof-of-work witness (the smallest
valid one).  Other shapes are checked by the reference's own in-circuit verifier semantics: the
verifier circuit built for that shape accepts the proof (witness generation without conflicts)."""
import json

import numpy as np
import pytest

from conftest import P
import p3json


def test_reproduces_reference_artifact_bit_for_bit(p25, fib_inputs):
    inp, cfg = p25.p3_prove_fibonacci(6, 100, 16)
    assert inp.shape == fib_inputs.shape
    assert (inp == fib_inputs).all()
    assert int(inp[8 + 16 + 24 + 9600 + 2]) == 103884            # pow_witness
    assert (cfg.log_trace_height, cfg.trace_width, cfg.degree_bits, cfg.num_queries) == (6, 3, 6, 100)


def test_json_roundtrip(p25, fib_inputs):
    inp, cfg = p25.p3_prove_fibonacci(5, 7, 6)
    text = p25.p3_inputs_to_json(inp, cfg)
    back, cfg2 = p25.p3_proof_from_json(text)
    assert (back == inp).all()
    assert (cfg2.log_trace_height, cfg2.num_queries, cfg2.degree_bits) == (5, 7, 5)
    obj = json.loads(text)
    assert (p3json.flatten_p3_proof(obj) == inp).all()            # independent Python reader agrees
    # the artifact re-serialised has the same element sequence
    text6 = p25.p3_inputs_to_json(fib_inputs, p25.P3Config.fib64())
    assert (p25.p3_proof_from_json(text6)[0] == fib_inputs).all()


@pytest.mark.parametrize("log_n,queries,pow_bits,pow_start", [(3, 4, 8, 0), (4, 10, 8, 0), (1, 3, 4, 0), (6, 5, 10, 77)])
def test_other_shapes_accepted_by_verifier_circuit(p25, oracle, log_n, queries, pow_bits, pow_start):
    inp, cfg = p25.p3_prove_fibonacci(log_n, queries, pow_bits, pow_start)
    c = p25.Circuit.build_p3_verifier(cfg)
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=1)
    assert st == 0, msg                       # every connect of src/p3 holds
    bad, msg = oc.check_constraints(wires)
    assert bad == 0, msg
    proof, st, _tm, msg = oc.prove(inp, seed=1)
    assert st == 0, msg
    assert oc.verify(proof)[0] == 0
    for pos in (0, len(inp) // 2, len(inp) - 1):
        t = inp.copy()
        t[pos] = (int(t[pos]) + 1) % P
        assert oc.witness(t, seed=1)[1] == 4


def test_distinct_pow_witness_gives_distinct_valid_proof(p25, oracle, fib_oracle, fib_inputs):
    """A second fib-64 proof (next valid PoW witness -> other query indices): accepted by the SAME
    circuit as the artifact, i.e. a genuinely different batch item for the fib-64 verifier circuit."""
    inp, _cfg = p25.p3_prove_fibonacci(6, 100, 16, pow_start=103885)
    assert int(inp[8 + 16 + 24 + 9600 + 2]) > 103884
    assert (inp != fib_inputs).sum() > 1000
    wires, st, msg = fib_oracle.witness(inp, seed=0)
    assert st == 0, msg
    assert fib_oracle.check_constraints(wires)[0] == 0


def test_bad_parameters(p25):
    for args in ((0, 10, 8), (23, 10, 8), (5, 0, 8)):
        with pytest.raises(p25.P25Error):
            p25.p3_prove_fibonacci(*args)
