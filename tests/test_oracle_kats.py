"""CPU: pin the oracle against every golden vector the reference tree holds for the hash path."""
import json
import os

import numpy as np

from conftest import ROOT, P

GOLD = os.path.join(ROOT, "tests", "golden")


def test_poseidon_v1_kats(oracle):
    # /root/reference/src/common/poseidon2/poseidon2_goldilocks.rs:190-211 (Poseidon-v1 vectors)
    g = json.load(open(os.path.join(GOLD, "poseidon_v1_constants.json")))
    assert len(g["kats"]) == 4
    for kat in g["kats"]:
        out = oracle.poseidon_permute(np.array(kat["input"], dtype=np.uint64))[0]
        assert [int(x) for x in out] == kat["output"]


def test_tuned_avx512_poseidon_equals_the_scalar_oracle(oracle):
    """oracle/ref_hash_x8.cpp (the tuned cpu_baseline leg of bench.py): eight permutations per call, bit-identical to the
    scalar oracle on the reference's four known-answer vectors (poseidon2_goldilocks.rs:190-211), on edge words and on
    random states; Merkle commitments of an LDE are the same with the switch on and off."""
    import pytest
    from conftest import splitmix_field
    if not oracle.x8_available():
        pytest.skip("no AVX-512 on this host")
    g = json.load(open(os.path.join(GOLD, "poseidon_v1_constants.json")))
    st = np.zeros((8, 12), dtype=np.uint64)
    for k, kat in enumerate(g["kats"]):
        st[k] = st[k + 4] = np.array(kat["input"], dtype=np.uint64)
    out = oracle.poseidon_permute_x8(st)
    for k, kat in enumerate(g["kats"]):
        assert [int(x) for x in out[k]] == kat["output"] and [int(x) for x in out[k + 4]] == kat["output"]
    edge = np.zeros((8, 12), dtype=np.uint64)
    edge[1, :], edge[2, :], edge[3, :], edge[4, :] = P - 1, 0xFFFFFFFF, 1 << 32, 0xFFFFFFFF00000000
    edge[5], edge[6], edge[7] = np.arange(12), P - 1 - np.arange(12, dtype=np.uint64), splitmix_field(12, seed=1)
    assert (oracle.poseidon_permute_x8(edge) == oracle.poseidon_permute(edge)).all()
    for seed in range(20):
        s = splitmix_field(96, seed=100 + seed).reshape(8, 12)
        assert (oracle.poseidon_permute_x8(s) == oracle.poseidon_permute(s)).all()
    # widths on both sides of the rate: 5 (one permutation), 8, 9, 20, 135 words per leaf
    try:
        for width in (5, 8, 9, 20, 135):
            vals = splitmix_field(width * 64, seed=width).reshape(width, 64)
            assert not oracle.set_tuned(False)
            plain = oracle.lde_commit(vals, 3, 4)
            assert oracle.set_tuned(True)
            tuned = oracle.lde_commit(vals, 3, 4)
            assert all((a == b).all() for a, b in zip(plain, tuned)), width
    finally:
        oracle.set_tuned(False)


def test_poseidon2_probe_vectors(oracle):
    # Poseidon2 outputs that reproduce the artifact's Merkle roots
    z = oracle.poseidon2_permute(np.zeros(12, dtype=np.uint64))[0]
    assert [hex(int(x)) for x in z[:4]] == ["0xb7c3a0ee7dfdcedf", "0xc4b98abeafdd334b",
                                            "0xd1334378971a1feb", "0x95d6c2ba775d318e"]
    r = oracle.poseidon2_permute(np.arange(12, dtype=np.uint64))[0]
    assert [hex(int(x)) for x in r[:4]] == ["0x6fee448d5ff51777", "0x534ad26248c82e0b",
                                            "0x9f455f73232afd26", "0xdb092454f2e6bde5"]


def test_poseidon2_trace_consistent(oracle):
    s0 = np.arange(12, dtype=np.uint64) * np.uint64(7919)
    out, tr = oracle.poseidon2_trace(s0)
    assert (out == oracle.poseidon2_permute(s0)[0]).all()
    assert (tr < np.uint64(P)).all() and tr.any()


def test_field_mul_inv(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = int(rng.integers(1, P, dtype=np.uint64))
        b = int(rng.integers(0, P, dtype=np.uint64))
        assert oracle.lib.p25o_mul(a, b) == (a * b) % P
        assert (oracle.lib.p25o_inv(a) * a) % P == 1
    assert oracle.lib.p25o_mul(P - 1, P - 1) == 1


def test_sponge_and_merkle_small(oracle):
    # hash_or_noop pads <= 4 words; tree of 8 leaves with cap height 1
    leaves = np.arange(8 * 3, dtype=np.uint64).reshape(8, 3)
    cap, tree = oracle.merkle_commit(leaves, 1, want_tree=True)
    assert tree[:4].tolist() == [0, 1, 2, 0]  # noop-padded leaf digest
    assert cap.shape == (2, 4)
    wide = np.arange(8 * 9, dtype=np.uint64).reshape(8, 9)
    cap2, tree2 = oracle.merkle_commit(wide, 0, want_tree=True)
    assert (tree2[:4] == oracle.hash_no_pad(wide[0])).all()
    assert cap2.shape == (1, 4)


def test_ntt_roundtrip_and_lde(oracle):
    rng = np.random.default_rng(2)
    vals = rng.integers(0, P, size=(3, 64), dtype=np.uint64)
    coeffs, lde, cap = oracle.lde_commit(vals, 3, 2)
    # LDE restricted to the subgroup coset: evaluate coefficient form directly at 7*w^i
    w = pow(7, (P - 1) // 512, P)
    for p in range(3):
        c = [int(x) for x in coeffs[p]]
        for i in (0, 1, 5, 511):
            x = 7 * pow(w, i, P) % P
            acc = 0
            for ck in reversed(c):
                acc = (acc * x + ck) % P
            r = int(format(i, "09b")[::-1], 2)
            assert int(lde[p][r]) == acc
        # ifft correctness: evaluating at w8^(8i) = subgroup of size 64
        w64 = pow(w, 8, P)
        for i in (0, 3, 63):
            x = pow(w64, i, P)
            acc = 0
            for ck in reversed(c):
                acc = (acc * x + ck) % P
            assert acc == int(vals[p][i])


def test_poseidon_fast_form_equals_definition(oracle):
    """The oracle's timed permutation (optimised partial rounds, as upstream's CPU code) equals the
    naive definition on random states; both are pinned by the KATs above."""
    import ctypes as C
    oracle.lib.p25o_poseidon_permute_naive.argtypes = [C.c_void_p, C.c_size_t]
    from conftest import splitmix_field
    s = splitmix_field(12 * 4000, seed=99).reshape(-1, 12)
    s[0, :] = 0
    s[1, :] = P - 1
    a = oracle.poseidon_permute(s)
    b = s.copy()
    oracle.lib.p25o_poseidon_permute_naive(b.ctypes.data, len(b))
    assert (a == b).all()


def test_poseidon_gate_wires_same_in_fast_and_naive_form(oracle):
    """upstream's PoseidonGate evaluates (and its generator fills) the partial rounds in the optimised form; this
    library's PoseidonGate uses the defining form.  The S-box inputs -- the gate's wires -- are the same values in
    both, so witnesses and constraint polynomials agree."""
    from conftest import splitmix_field
    for seed in range(5):
        st = splitmix_field(12, seed=900 + seed)
        out_n, tr = oracle.poseidon_trace(st)
        out_f, pin = oracle.poseidon_fast_partial_inputs(st)
        assert (out_n == out_f).all() and (out_n == oracle.poseidon_permute(st)[0]).all()
        assert (tr[36:58] == pin).all()


def test_avx512_code_stays_in_the_x8_functions():
    """oracle/ref_quotient_x8.cpp is compiled for AVX-512 as a whole and includes the shared oracle headers; an inline or
    template helper it instantiated must never be the copy the scalar checker runs (SIGILL on a host without AVX-512, with the
    tuned leg off).  Every function of libp25_oracle.so that touches a zmm register must be one of the eight-lane ones."""
    import re
    import shutil
    import subprocess
    objdump = shutil.which("objdump")
    if not objdump:
        pytest.skip("objdump not available")
    lib = os.path.join(ROOT, "oracle", "libp25_oracle.so")
    out = subprocess.run([objdump, "-d", "--no-show-raw-insn", "-C", lib], capture_output=True, text=True, timeout=600).stdout
    fn, offenders = None, set()
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            fn = m.group(1)
        elif "zmm" in line and fn is not None and not re.search(r"FB8|_x8|^v_(poseidon|mds)\b", fn):
            offenders.add(fn)
    assert not offenders, sorted(offenders)[:10]
