"""GPU parity on verifier circuits for other plonky3 proof shapes (inputs from the native p3 prover):
every circuit size from 2^11 to 2^18 rows (2^16 and 2^19 are the fib-64 and config-5 tests), i.e. every NTT split between the
single-pass 2^10 and the 2^9 x 2^10 of config 5, FRI schedules of two to four layers and final polynomials of 2^3 .. 2^6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,queries,pow_bits", [(3, 4, 8), (4, 10, 8), (6, 12, 8), (6, 25, 8), (6, 50, 8), (8, 100, 16),
                                                    (11, 100, 16)])   # outer rows 2^11, 2^12, 2^13, 2^14, 2^15, 2^17, 2^18
def test_gpu_equals_oracle_on_shape(gpu, oracle, log_n, queries, pow_bits):
    inp, cfg = gpu.p3_prove_fibonacci(log_n, queries, pow_bits)
    c = gpu.Circuit.build_p3_verifier(cfg)
    oc = oracle.load_circuit(c.to_blob())
    dg, capg = c.digest()
    do, capo = oc.digest()
    assert (dg == do).all() and (capg == capo).all()
    other, _ = gpu.p3_prove_fibonacci(log_n, queries, pow_bits, pow_start=int(inp[-1 - 0]) if False else 1 << 20)
    proofs, st = c.prove(np.stack([inp, other]), seeds=[3, 4])
    assert st.tolist() == [0, 0]
    po, sto, _tm, msg = oc.prove(inp, seed=3)
    assert sto == 0, msg
    diff = np.nonzero(proofs[0] != po)[0]
    assert diff.size == 0, diff[:8]
    assert oc.verify(proofs[1], dg, capg)[0] == 0


def test_two_distinct_fib64_proofs_in_one_batch(gpu, fib_circuit, fib_oracle, fib_inputs):
    inp2, _ = gpu.p3_prove_fibonacci(6, 100, 16, pow_start=103885)
    proofs, st = fib_circuit.prove(np.stack([fib_inputs, inp2]), seeds=[1, 1])
    assert st.tolist() == [0, 0]
    assert (proofs[0] != proofs[1]).any()
    dg, capg = fib_circuit.digest()
    for p in proofs:
        assert fib_oracle.verify(p, dg, capg)[0] == 0


@pytest.mark.parametrize("name,log_n", [("tribonacci", 4), ("squares", 5), ("cubic", 5), ("cubic_transition", 4)])
def test_gpu_equals_oracle_on_user_air(gpu, oracle, name, log_n):
    """verifier circuits for AIRs given as data (p25_air) -- width 4 / a quadratic and a
    last-row constraint / constraint degree 3, i.e. TWO quotient chunks (round 5: the chunk count of serde/proof.rs:41-48
    lifted; the reference's verifier.rs handles any power of two) -- proved on the GPU, byte-identical to the oracle; a
    tampered input fails."""
    import air_cases
    air = getattr(air_cases, name)(gpu)
    inp, cfg = gpu.p3_prove_air(air, getattr(air_cases, name + "_trace")(log_n), num_queries=12, pow_bits=8)
    c = gpu.Circuit.build_p3_verifier_air(cfg, air)
    oc = oracle.load_circuit(c.to_blob())
    dg, capg = c.digest()
    do, capo = oc.digest()
    assert (dg == do).all() and (capg == capo).all()
    bad = inp.copy()
    bad[9] = (int(bad[9]) + 1) % 0xFFFFFFFF00000001
    proofs, st = c.prove(np.stack([inp, bad]), seeds=[5, 6])
    assert st.tolist() == [0, 4]
    po, sto, _tm, msg = oc.prove(inp, seed=5)
    assert sto == 0, msg
    assert (proofs[0] == po).all()
    assert oc.verify(proofs[0], dg, capg)[0] == 0


@pytest.mark.parametrize("family,seed,arg", [("random_recurrence", 1, 2), ("random_recurrence", 2, 5), ("random_recurrence", 3, 9),
                                             ("quadratic_pair", 11, None), ("quadratic_pair", 12, None)])
def test_gpu_equals_oracle_on_seeded_air_families(gpu, oracle, family, seed, arg):
    """SURVEY.md 8f-2 beyond hand-picked AIRs: seeded families -- linear recurrences of widths 2, 5, 9 with random
    coefficients, pairs of quadratic always-constraints with random constants -- through the native plonky3 prover, the
    verifier-circuit builder and the GPU prover: digest and proof bytes equal the oracle's, a tampered input has no witness."""
    import air_cases
    if family == "random_recurrence":
        air, coef = air_cases.random_recurrence(gpu, seed, arg)
        trace = air_cases.random_recurrence_trace(coef, 4)
    else:
        air, par = air_cases.quadratic_pair(gpu, seed)
        trace = air_cases.quadratic_pair_trace(par, 5)
    inp, cfg = gpu.p3_prove_air(air, trace, num_queries=10, pow_bits=6)
    c = gpu.Circuit.build_p3_verifier_air(cfg, air)
    oc = oracle.load_circuit(c.to_blob())
    dg, capg = c.digest()
    do, capo = oc.digest()
    assert (dg == do).all() and (capg == capo).all()
    bad = inp.copy()
    bad[inp.size // 2] = (int(bad[inp.size // 2]) + 1) % 0xFFFFFFFF00000001
    proofs, st = c.prove(np.stack([inp, bad]), seeds=[5, 6])
    assert st[0] == 0 and st[1] != 0
    po, sto, _tm, msg = oc.prove(inp, seed=5)
    assert sto == 0, msg
    assert (proofs[0] == po).all()
    assert oc.verify(proofs[0], dg, capg)[0] == 0


@pytest.mark.parametrize("family,seed,log_n,log_blowup", [("quartic_map", 21, 4, 2), ("quartic_map", 23, 5, 3),
                                                          ("quintic_selector", 31, 5, 2), ("quintic_selector", 33, 3, 2)])
def test_gpu_equals_oracle_on_degree_four_and_five_airs_with_log_blowup_two(gpu, oracle, family, seed, log_n, log_blowup):
    """FriConfig.log_blowup 2 / 3 (src/p3/mod.rs:242-246) and AIRs of constraint degree 4 and 5 -- FOUR quotient chunks --
    through `p25_p3_prove_air_ex` -> `p25_circuit_build_p3_verifier_air` -> the GPU prover: circuit digest and proof bytes
    equal the oracle's, the oracle's verifier accepts, and a flipped input (one of the chunk openings, and a word in the
    middle of the query openings) has no witness on either side."""
    import air_cases
    air, par = getattr(air_cases, family)(gpu, seed)
    trace = getattr(air_cases, family + "_trace")(par, log_n)
    inp, cfg = gpu.p3_prove_air(air, trace, num_queries=8, pow_bits=6, log_blowup=log_blowup)
    assert (cfg.log_blowup, cfg.log_quotient_degree) == (log_blowup, 2)
    alt, _ = gpu.p3_prove_air(air, trace, num_queries=8, pow_bits=6, pow_start=1 << 20, log_blowup=log_blowup)
    c = gpu.Circuit.build_p3_verifier_air(cfg, air)
    oc = oracle.load_circuit(c.to_blob())
    dg, capg = c.digest()
    do, capo = oc.digest()
    assert (dg == do).all() and (capg == capo).all()
    bad1, bad2 = inp.copy(), inp.copy()
    k1, k2 = 8 + 4 * air.width + 5, (2 * inp.size) // 3
    bad1[k1] = (int(bad1[k1]) + 1) % 0xFFFFFFFF00000001
    bad2[k2] = (int(bad2[k2]) + 1) % 0xFFFFFFFF00000001
    proofs, st = c.prove(np.stack([inp, bad1, alt, bad2]), seeds=[5, 6, 7, 8])
    assert st.tolist() == [0, 4, 0, 4]
    for row, seed_ in ((inp, 5), (alt, 7)):
        po, sto, _tm, msg = oc.prove(row, seed=seed_)
        assert sto == 0, msg
        assert (proofs[0 if seed_ == 5 else 2] == po).all()
    assert oc.witness(bad1, seed=6)[1] == 4 and oc.witness(bad2, seed=8)[1] == 4
    assert oc.verify(proofs[0], dg, capg)[0] == 0 and oc.verify(proofs[2], dg, capg)[0] == 0
