"""GPU: the prover's stages on their own, through the fine-grained C-ABI entry points, against the
oracle's restatement of the same upstream functions -- so a regression in a6-a10 is reported at its stage instead
of as "first differing proof word".

  a6  Challenger                      p25_transcript        vs RChallenger
  a7  partial products + Z            p25_partial_products  vs ref_partial_products
  a8  quotient + every gate evaluator p25_quotient          vs ref_quotient_chunks
  a9  openings f(zeta), Z(g zeta)     p25_eval_polys        vs Horner in F_p^2
  a10 FRI commit / PoW / queries      p25_fri_prove         vs ref_fri_prove
"""
import numpy as np
import pytest

from conftest import P, splitmix_field

pytestmark = pytest.mark.gpu


def test_transcript_scripts_vs_oracle(gpu, oracle):
    rng = np.random.default_rng(7)
    for trial in range(12):
        segs = []
        for _ in range(int(rng.integers(1, 7))):
            n_obs = int(rng.choice([0, 1, 3, 7, 8, 9, 15, 16, 17, 64, 100, 327]))
            n_ch = int(rng.choice([0, 1, 2, 4, 7, 8, 9, 29]))
            segs.append((splitmix_field(n_obs, seed=int(rng.integers(1, 1 << 60))), n_ch))
        g = gpu.transcript(segs)
        o = oracle.transcript(segs)
        assert (g == o).all(), (trial, [(len(w), k) for w, k in segs])
        assert (g < np.uint64(P)).all()
    # draw without observing anything; observe-only scripts
    assert (gpu.transcript([([], 3)]) == oracle.transcript([([], 3)])).all()
    assert gpu.transcript([(splitmix_field(5), 0)]).size == 0


@pytest.fixture(scope="module")
def small(gpu, oracle):
    """plonky3-verifier circuit of a 2^3-row Fibonacci STARK (2^10 rows, every gate type of the fib-64 circuit)."""
    inp, cfg = gpu.p3_prove_fibonacci(3, 3, 4)
    c = gpu.Circuit.build_p3_verifier(cfg)
    oc = oracle.load_circuit(c.to_blob())
    wires, st, msg = oc.witness(inp, seed=3)
    assert st == 0, msg
    return c, oc, wires


def _chal(seed):
    return splitmix_field(6, seed=seed).reshape(3, 2)


def _check_stage_a7_a8(c, oc, wires, seed):
    betas, gammas, alphas = _chal(seed)
    zg = c.partial_products(wires, betas, gammas)
    zo = oc.partial_products(wires, betas, gammas)
    assert zg.shape == zo.shape and (zg == zo).all(), np.argwhere(zg != zo)[:5]
    assert (zg[:2, 0] == 1).all()                       # Z(1) = 1
    qg = c.quotient(wires, zg, betas, gammas, alphas)
    qo = oc.quotient(wires, zo, betas, gammas, alphas)
    assert qg.shape == qo.shape and (qg == qo).all(), np.argwhere(qg != qo)[:5]
    return zg, qg


def test_partial_products_and_quotient_vs_oracle_small(small):
    c, oc, wires = small
    for seed in (11, 12):
        _zs, q = _check_stage_a7_a8(c, oc, wires, seed)
        # a satisfied witness: the quotient is a polynomial of degree < 8n whose top chunk stays below the gate degree
        assert q.any()


def test_quotient_sees_every_wire_column(small):
    """Changing one witness value changes the quotient chunks on both sides in the same way (the evaluator of the
    row's gate reads that column), and the result no longer has the low degree of a valid quotient."""
    c, oc, wires = small
    betas, gammas, alphas = _chal(5)
    zs = oc.partial_products(wires, betas, gammas)
    base = oc.quotient(wires, zs, betas, gammas, alphas)
    bad = wires.copy()
    bad[1, 0] = (int(bad[1, 0]) + 1) % P
    qg = c.quotient(bad, zs, betas, gammas, alphas)
    qo = oc.quotient(bad, zs, betas, gammas, alphas)
    assert (qg == qo).all() and (qg != base).any()


def test_partial_products_and_quotient_vs_oracle_fib64(gpu, fib_circuit, fib_oracle, fib_inputs):
    """Full size: 2^16 rows x 135 wires, LDE 2^19 -- the shapes k_zpp_* and k_quotient run at in the bench."""
    wires, st, msg = fib_oracle.witness(fib_inputs, seed=77)
    assert st == 0, msg
    _check_stage_a7_a8(fib_circuit, fib_oracle, wires, 21)


@pytest.mark.parametrize("log_n,rate_bits,cap_h,arity,pow_bits,queries", [
    (10, 3, 4, [4, 4], 8, 5),          # two layers, as a 2^10-row circuit has
    (12, 3, 4, [4, 4], 10, 28),        # 28 queries, final polynomial of 2^4 coefficients
    (12, 3, 2, [4, 4, 4], 10, 9),      # folds down to a constant final polynomial
    (8, 1, 0, [3, 2, 1], 4, 7),        # ragged arities, cap of one digest
    (6, 2, 2, [], 6, 3),               # no commit-phase layer at all: final polynomial = the input
    (16, 3, 4, [4, 4, 4], 16, 28),     # the fib-64 circuit's FRI: 2^16 coefficients, LDE 2^19, 16 PoW bits
])
def test_fri_prove_vs_oracle(gpu, oracle, log_n, rate_bits, cap_h, arity, pow_bits, queries):
    coeffs = splitmix_field(2 << log_n, seed=1000 + log_n).reshape(2, 1 << log_n)
    seed = splitmix_field(13, seed=99)
    g, st = gpu.fri_prove(coeffs, rate_bits, cap_h, arity, pow_bits, queries, seed)
    assert st == 0
    o = oracle.fri_prove(coeffs, rate_bits, cap_h, arity, pow_bits, queries, seed)
    assert g.shape == o.shape
    diff = np.nonzero(g != o)[0]
    assert diff.size == 0, f"first differing words {diff[:8]} of {g.size}"


def test_fri_bad_shapes_rejected(gpu, p25):
    coeffs = splitmix_field(2 << 6).reshape(2, 64)
    for arity in ([7], [4, 4], [0], [9]):                 # folds below degree 1 / below the cap / out of range
        with pytest.raises(p25.P25Error):
            gpu.fri_prove(coeffs, 1, 2, arity, 4, 3, [1, 2, 3])
    bad = coeffs.copy()
    bad[0, 0] = np.uint64(P)
    with pytest.raises(p25.P25Error):
        gpu.fri_prove(bad, 1, 2, [2], 4, 3, [1, 2, 3])


@pytest.mark.parametrize("log_n,n_polys", [(4, 3), (10, 7), (16, 20), (18, 2)])
def test_openings_vs_oracle(gpu, oracle, log_n, n_polys):
    """a9: polynomial openings at an extension point -- the strided-Horner + LDS reduction kernel, including the chunked
    form for polynomials longer than 2^16 (config 5) and the scaled point g*zeta of the next-row openings."""
    coeffs = splitmix_field(n_polys << log_n, seed=70 + log_n).reshape(n_polys, 1 << log_n)
    zeta = splitmix_field(2, seed=71)
    g = pow(1753635133440165772, 1 << (32 - log_n), P)
    for scale in (1, g):
        got = gpu.eval_polys(coeffs, zeta, scale)
        want = oracle.eval_polys(coeffs, zeta, scale)
        assert (got == want).all(), (log_n, scale)


def test_mark_and_wait_mark_argument_checks(gpu, small):
    """The event-ordering entry points refuse what cannot be right (status codes, never UB): slots beyond 0..15 (P25_MAX_MARKS), a circuit
    waiting for its own mark (its streams are already in order), null handles."""
    c, _oc, _wires = small
    other = gpu.Circuit.build_gadget(0, 0)
    c.mark(15)                                  # the last valid slot
    for slot in (16, 1 << 20):
        with pytest.raises(gpu.P25Error) as e:
            c.mark(slot)
        assert e.value.status == 1
        with pytest.raises(gpu.P25Error):
            other.wait_mark(c, slot)
        with pytest.raises(gpu.P25Error):
            c.stream_wait_mark(slot, 0)
    with pytest.raises(gpu.P25Error) as e:
        c.wait_mark(c, 0)
    assert e.value.status == 1
    lib = gpu.lib()
    assert lib.p25_circuit_mark(None, 0) == 1 and lib.p25_circuit_wait_mark(None, None, 0) == 1
    assert lib.p25_circuit_stream_join(None, None) == 1 and lib.p25_circuit_wait_stream(None, None) == 1
    # a mark nobody has taken yet is nothing to wait for; marks and waits on the legacy default stream are accepted
    other.wait_mark(c, 3)
    c.mark(3)
    other.wait_mark(c, 3)
    c.stream_wait_mark(3, 0)
    c.stream_join(0)
    c.wait_stream(0)
    c.sync()
    other.sync()
    other.close()
