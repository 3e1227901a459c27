"""CPU: AIRs as data -- `p25_circuit_build_p3_verifier_air`, `p25_p3_prove_air`.

Pins: the FibonacciAir of src/p3/mod.rs:176-221 written as a program must give (a) the SAME circuit as
the hand-written restatement (same digest, computed by the oracle) and (b) the SAME proof as the
Fibonacci prover, i.e. the reference's artifact bit for bit.  Other AIRs are checked through the
reference's own in-circuit verifier semantics: the verifier circuit built for that AIR accepts the
native proof (every `connect` of src/p3 holds), rejects tampered proofs, and a trace that violates the
AIR cannot be proved."""
import numpy as np
import pytest

import air_cases
from conftest import P


def test_fibonacci_program_builds_the_same_circuit(p25, oracle, fib_blob):
    c = p25.Circuit.build_p3_verifier_air(p25.P3Config.fib64(), p25.Air.fibonacci())
    assert c.to_blob() == fib_blob


def test_fibonacci_program_reproduces_the_artifact(p25, fib_inputs):
    inp, cfg = p25.p3_prove_air(p25.Air.fibonacci(), air_cases.fib_trace(6), 100, 16)
    assert (inp == fib_inputs).all()
    assert (cfg.trace_width, cfg.log_trace_height) == (3, 6)


@pytest.mark.parametrize("name,log_n", [("tribonacci", 4), ("squares", 3), ("squares", 5)])
def test_user_air_accepted_by_its_verifier_circuit(p25, oracle, name, log_n):
    air = getattr(air_cases, name)(p25)
    trace = getattr(air_cases, name + "_trace")(log_n)
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=6, pow_bits=6)
    assert cfg.trace_width == air.width
    c = p25.Circuit.build_p3_verifier_air(cfg, air)
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=3)
    assert st == 0, msg
    bad, msg = oc.check_constraints(wires)
    assert bad == 0, msg
    proof, st, _tm, msg = oc.prove(inp, seed=3)
    assert st == 0, msg
    assert oc.verify(proof)[0] == 0
    # tampering with an opened trace value or the quotient opening breaks a `connect`
    for pos in (8, 8 + 4 * air.width, len(inp) - 1):
        t = inp.copy()
        t[pos] = (int(t[pos]) + 1) % P
        assert oc.witness(t, seed=3)[1] == 4
    # the Fibonacci verifier circuit of the same shape does not accept this AIR's proof
    if air.width == 3:
        fc = oracle.load_circuit(p25.Circuit.build_p3_verifier(cfg).to_blob())
        assert fc.witness(inp, seed=3)[1] == 4


@pytest.mark.parametrize("family,seed,arg", [("random_recurrence", 1, 2), ("random_recurrence", 2, 5), ("random_recurrence", 4, 12),
                                             ("quadratic_pair", 11, None), ("quadratic_pair", 13, None)])
def test_seeded_air_families_accepted_by_their_verifier_circuits(p25, oracle, family, seed, arg):
    """Beyond hand-picked AIRs: seeded families (tests/air_cases.py) through the native plonky3 prover and the
    verifier-circuit builder; the circuit's witness exists and satisfies every constraint, a violating trace cannot be
    proved, a tampered proof has no witness."""
    if family == "random_recurrence":
        air, coef = air_cases.random_recurrence(p25, seed, arg)
        trace = air_cases.random_recurrence_trace(coef, 4)
    else:
        air, par = air_cases.quadratic_pair(p25, seed)
        trace = air_cases.quadratic_pair_trace(par, 4)
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=5, pow_bits=4)
    c = p25.Circuit.build_p3_verifier_air(cfg, air)
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=2)
    assert st == 0, msg
    bad, msg = oc.check_constraints(wires)
    assert bad == 0, msg
    t = inp.copy()
    t[8] = (int(t[8]) + 1) % P
    assert oc.witness(t, seed=2)[1] == 4
    wrong = trace.copy()
    wrong[3, air.width - 1] = (int(wrong[3, air.width - 1]) + 1) % P
    with pytest.raises(p25.P25Error):
        p25.p3_prove_air(air, wrong, num_queries=5, pow_bits=4)


@pytest.mark.parametrize("name,log_n", [("cubic", 4), ("cubic", 6), ("cubic_transition", 3), ("cubic_transition", 5)])
def test_degree_three_air_two_quotient_chunks(p25, oracle, name, log_n):
    """Constraint degree 3 (in the constraint itself, or a quadratic one under a selector): TWO quotient chunks.  The
    reference's verifier handles any power of two (verifier.rs:115-221: split domains, zps, recomposition; mod.rs:76 derives
    log_quotient_degree from the proof) and only its proof model fixes one (serde/proof.rs:41-48 `(0..1)`); with that lifted
    -- one more chunk's openings after the first, one more matrix in the quotient batch -- the native prover's proof is
    accepted by the verifier circuit built for the AIR (witness exists, every constraint zero, the plonky2 proof of it
    verifies), any flipped input word is rejected, the JSON form round-trips, and the shapes cannot be mixed up."""
    import p3json, json
    air = getattr(air_cases, name)(p25)
    trace = getattr(air_cases, name + "_trace")(log_n)
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=6, pow_bits=6)
    assert cfg.log_quotient_degree == 1 and cfg.trace_width == air.width
    c = p25.Circuit.build_p3_verifier_air(cfg, air)
    assert int(c.info.num_inputs) == inp.size
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=3)
    assert st == 0, msg
    bad, msg = oc.check_constraints(wires)
    assert bad == 0, msg
    proof, st, _tm, msg = oc.prove(inp, seed=3)
    assert st == 0, msg
    assert oc.verify(proof)[0] == 0
    # both chunks' openings (4 + 4 words behind the trace openings), commitments, FRI and query words: no flip survives
    q0 = 8 + 4 * air.width
    for pos in list(range(q0, q0 + 8)) + [0, 4, 8, q0 + 8, len(inp) // 2, len(inp) - 5, len(inp) - 1]:
        t = inp.copy()
        t[pos] = (int(t[pos]) + 1) % P
        assert oc.witness(t, seed=3)[1] == 4, pos
    # the JSON form carries two chunks; the independent Python reader flattens it to the same vector and derives the shape
    js = p25.p3_inputs_to_json(inp, cfg)
    obj = json.loads(js)
    assert len(obj["opened_values"]["quotient_chunks"]) == 2
    assert len(obj["opening_proof"]["query_openings"][0][1]["opened_values"]) == 2
    assert (p3json.flatten_p3_proof(obj) == inp).all() and p3json.p3_shape(obj)["log_quotient_degree"] == 1
    back, cfg2 = p25.p3_proof_from_json(js)
    assert (back == inp).all() and cfg2.log_quotient_degree == 1
    # shapes cannot be mixed up: this AIR with a one-chunk shape, a degree-2 AIR with a two-chunk shape
    one = p25.P3Config(cfg.log_blowup, cfg.num_queries, cfg.proof_of_work_bits, 0, cfg.log_trace_height, cfg.trace_width,
                       cfg.opening_matrix_log_max_height, cfg.quotient_opened_len, cfg.degree_bits)
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(one, air)
    if air.width == 3:
        with pytest.raises(p25.P25Error):
            p25.Circuit.build_p3_verifier_air(cfg, p25.Air.fibonacci())
    # a trace that breaks the cubic relation cannot be proved
    wrong = trace.copy()
    wrong[2, air.width - 1] = (int(wrong[2, air.width - 1]) + 1) % P
    with pytest.raises(p25.P25Error):
        p25.p3_prove_air(air, wrong, num_queries=6, pow_bits=6)


@pytest.mark.parametrize("family,seed,log_n,log_blowup", [("quartic_map", 21, 4, 2), ("quartic_map", 22, 3, 3), ("quartic_map", 24, 2, 4),
                                                          ("quintic_selector", 31, 5, 2), ("quintic_selector", 32, 2, 2)])
def test_degree_four_and_five_airs_four_quotient_chunks_log_blowup_two(p25, oracle, family, seed, log_n, log_blowup):
    """FriConfig.log_blowup > 1 end to end (src/p3/mod.rs:242-246; verifier.rs:264, 299, 378, 397 read it generically): AIRs
    of constraint degree 4 (an ALWAYS constraint) and 5 (a quartic transition times its selector) have 2^2 quotient chunks,
    which the LDE domain 7*H_{4n} holds.  The native plonky3 prover (log_blowup 2 and 3), the shape, the JSON forms and the
    verifier circuit agree: the witness exists, every constraint of the outer circuit vanishes, the oracle proves and verifies
    it, and no flipped input word survives -- chunk openings, commitments, FRI layers, query openings."""
    import json
    import p3json
    air, par = getattr(air_cases, family)(p25, seed)
    trace = getattr(air_cases, family + "_trace")(par, log_n)
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=6, pow_bits=6, log_blowup=log_blowup)
    assert (cfg.log_blowup, cfg.log_quotient_degree, cfg.log_trace_height, cfg.opening_matrix_log_max_height) == \
        (log_blowup, 2, log_n, log_n + log_blowup)
    c = p25.Circuit.build_p3_verifier_air(cfg, air)
    assert int(c.info.num_inputs) == inp.size
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=3)
    assert st == 0, msg
    bad, msg = oc.check_constraints(wires)
    assert bad == 0, msg
    proof, st, _tm, msg = oc.prove(inp, seed=3)
    assert st == 0, msg
    assert oc.verify(proof)[0] == 0
    q0 = 8 + 4 * air.width                       # the four chunks' openings: 16 words behind the trace openings
    for pos in list(range(q0, q0 + 16)) + [0, 4, 8, q0 + 16, len(inp) // 3, len(inp) // 2, len(inp) - 5, len(inp) - 1]:
        t = inp.copy()
        t[pos] = (int(t[pos]) + 1) % P
        assert oc.witness(t, seed=3)[1] == 4, pos
    # another PoW witness: other query indices, same circuit
    alt, _ = p25.p3_prove_air(air, trace, num_queries=6, pow_bits=6, pow_start=1 << 20, log_blowup=log_blowup)
    assert (alt != inp).any() and oc.witness(alt, seed=4)[1] == 0
    # JSON: four chunks, one matrix per chunk in the quotient batch, Merkle paths of log_n + log_blowup (inputs) and
    # log_n + log_blowup - 1 - i (commit phase); both readers recover the vector and the shape, log_blowup included
    js = p25.p3_inputs_to_json(inp, cfg)
    obj = json.loads(js)
    assert len(obj["opened_values"]["quotient_chunks"]) == 4
    qo = obj["opening_proof"]["query_openings"][0]
    assert len(qo[1]["opened_values"]) == 4 and len(qo[0]["opening_proof"]) == log_n + log_blowup
    steps = obj["opening_proof"]["fri_proof"]["query_proofs"][0]["commit_phase_openings"]
    assert [len(s["opening_proof"]) for s in steps] == [log_n + log_blowup - 1 - i for i in range(log_n)]
    assert (p3json.flatten_p3_proof(obj) == inp).all()
    assert (p3json.p3_shape(obj)["log_quotient_degree"], p3json.p3_shape(obj)["log_blowup"]) == (2, log_blowup)
    back, cfg2 = p25.p3_proof_from_json(js)
    assert (back == inp).all() and (cfg2.log_quotient_degree, cfg2.log_blowup) == (2, log_blowup)
    # the shape must match the AIR and the FRI parameters
    for lqd, lb in ((1, log_blowup), (2, 1)):
        wrong = p25.P3Config(lb, cfg.num_queries, cfg.proof_of_work_bits, lqd, cfg.log_trace_height, cfg.trace_width,
                             cfg.log_trace_height + lb, cfg.quotient_opened_len, cfg.degree_bits)
        with pytest.raises(p25.P25Error):
            p25.Circuit.build_p3_verifier_air(wrong, air)
    with pytest.raises(p25.P25Error):            # four chunks do not fit log_blowup 1
        p25.p3_prove_air(air, trace, num_queries=6, pow_bits=6, log_blowup=1)
    wrong = trace.copy()
    wrong[1, air.width - 1] = (int(wrong[1, air.width - 1]) + 1) % P
    with pytest.raises(p25.P25Error):            # a trace that breaks the relation cannot be proved
        p25.p3_prove_air(air, wrong, num_queries=6, pow_bits=6, log_blowup=log_blowup)


def test_log_blowup_two_on_the_reference_air(p25, oracle):
    """The reference's own FibonacciAir (src/p3/mod.rs:176-221) under FriConfig.log_blowup = 2: one chunk, LDE 7*H_{4n}."""
    air = p25.Air.fibonacci()
    trace = air_cases.fib_trace(5)
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=5, pow_bits=5, log_blowup=2)
    assert (cfg.log_blowup, cfg.log_quotient_degree, cfg.opening_matrix_log_max_height) == (2, 0, 7)
    c = p25.Circuit.build_p3_verifier_air(cfg, air)
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=1)
    assert st == 0, msg
    assert oc.check_constraints(wires)[0] == 0
    ref, cfg1 = p25.p3_prove_air(air, trace, num_queries=5, pow_bits=5)       # the log_blowup-1 proof of the same trace
    assert cfg1.log_blowup == 1 and ref.size < inp.size
    t = inp.copy()
    t[inp.size - 3] = (int(t[inp.size - 3]) + 1) % P
    assert oc.witness(t, seed=1)[1] == 4


def test_violating_trace_cannot_be_proved(p25):
    air = air_cases.tribonacci(p25)
    trace = air_cases.tribonacci_trace(4)
    trace[5, 3] = (int(trace[5, 3]) + 1) % P
    with pytest.raises(p25.P25Error):
        p25.p3_prove_air(air, trace, num_queries=4, pow_bits=4)


def test_malformed_programs_rejected(p25):
    cfg = p25.P3Config.fib64()
    air = p25.Air(3)
    x = air.local(0)
    air.when_transition(air.sub(air.mul(x, x), air.local(1)))      # degree 3 with the selector: needs a two-chunk shape
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(cfg, air)
    air = p25.Air(3)
    x = air.local(0)
    air.assert_zero(air.sub(air.mul(air.mul(x, x), air.mul(x, x)), air.local(1)))   # degree 4: four chunks, not this shape's one
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(cfg, air)
    air = p25.Air(3)
    air.assert_zero(air.local(7))                                   # column out of range
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(cfg, air)
    air = p25.Air(3)
    air.nodes.append((3, 5, 6, 0))                                  # forward reference
    air.assert_zero(0)
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(cfg, air)
    air = p25.Air.fibonacci()
    cfg4 = p25.P3Config(1, 100, 16, 0, 6, 4, 7, 2, 6)               # width mismatch: "Invalid Proof Shape"
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(cfg4, air)
