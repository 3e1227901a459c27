"""ORACLE -- test infrastructure only: ctypes view of oracle/libp25_oracle.so (the CPU restatement).

Imported by tests/, __graft_entry__.smoke() and bench.py's verification / cpu_baseline legs -- never by
anything under plonky2.5_amd/.  The product has no CPU fallback and does not know this file exists."""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
P = 0xFFFFFFFF00000001


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleCircuit:
    """A circuit blob loaded into the oracle: CPU witness generator, prover and verifier."""

    def __init__(self, lib, blob):
        self.lib = lib
        self.h = C.c_void_p(lib.p25o_circuit_load(blob, len(blob)))
        assert self.h.value, "oracle could not parse the circuit blob"
        info = (C.c_uint64 * 8)()
        lib.p25o_circuit_info(self.h, info)
        self.degree_bits, self.num_wires, self.num_inputs, self.num_generators = map(int, info[:4])
        self.proof_words = int(info[4])
        self.n = 1 << self.degree_bits

    def witness(self, inputs, seed=0):
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        wires = np.zeros((self.num_wires, self.n), dtype=np.uint64)
        msg = C.create_string_buffer(512)
        st = self.lib.p25o_witness(self.h, _p(inp), seed, _p(wires), msg, 512)
        return wires, st, msg.value.decode()

    def witness_forced(self, inputs, seed=0):
        """witness() that carries on past a copy-constraint conflict (the partition keeps its first value)."""
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        wires = np.zeros((self.num_wires, self.n), dtype=np.uint64)
        msg = C.create_string_buffer(512)
        st = self.lib.p25o_witness_forced(self.h, _p(inp), seed, _p(wires), msg, 512)
        return wires, st, msg.value.decode()

    def check_constraints(self, wires):
        msg = C.create_string_buffer(512)
        bad = self.lib.p25o_check_constraints(self.h, _p(np.ascontiguousarray(wires)), msg, 512)
        return bad, msg.value.decode()

    def digest(self):
        d = np.zeros(4, dtype=np.uint64)
        cap = np.zeros((16, 4), dtype=np.uint64)
        self.lib.p25o_circuit_digest(self.h, _p(d), _p(cap))
        return d, cap

    def prove(self, inputs, seed=0):
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        proof = np.zeros(self.proof_words, dtype=np.uint64)
        tm = (C.c_double * 9)()
        msg = C.create_string_buffer(512)
        st = self.lib.p25o_prove(self.h, _p(inp), seed, _p(proof), tm, msg, 512)
        names = ["witness", "wires_commit", "zs", "zs_commit", "quotient", "quotient_commit", "openings", "fri", "total"]
        return proof, st, dict(zip(names, [float(x) for x in tm])), msg.value.decode()

    def prove_filler(self, inputs, filler):
        """Prove with explicit RandomValueGenerator values (what a real upstream run drew from the OS RNG)."""
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        f = np.ascontiguousarray(filler, dtype=np.uint64)
        assert f.size == self.lib.p25o_num_random_fill(self.h)
        proof = np.zeros(self.proof_words, dtype=np.uint64)
        msg = C.create_string_buffer(512)
        st = self.lib.p25o_prove_filler(self.h, _p(inp), _p(f), _p(proof), msg, 512)
        return proof, st, msg.value.decode()

    def partial_products(self, wires, betas, gammas):
        w = np.ascontiguousarray(wires, dtype=np.uint64)
        nz = self._nz()
        out = np.zeros((nz, self.n), dtype=np.uint64)
        self.lib.p25o_partial_products(self.h, _p(w), _p(np.ascontiguousarray(betas, dtype=np.uint64)),
                                       _p(np.ascontiguousarray(gammas, dtype=np.uint64)), _p(out))
        return out

    def _nz(self):
        nz = C.c_uint64(0)
        self.lib.p25o_stage_shapes(self.h, C.byref(nz), None)
        return int(nz.value)

    def quotient(self, wires, zs_pp, betas, gammas, alphas):
        nq = C.c_uint64(0)
        self.lib.p25o_stage_shapes(self.h, None, C.byref(nq))
        out = np.zeros((int(nq.value), self.n), dtype=np.uint64)
        u = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
        ww, zz, b, g, al = u(wires), u(zs_pp), u(betas), u(gammas), u(alphas)
        self.lib.p25o_quotient(self.h, _p(ww), _p(zz), _p(b), _p(g), _p(al), _p(out))
        return out

    def prove_many(self, inputs, seeds, threads, want_proofs=True, threads_per_proof=1):
        """N independent proofs, `threads` in flight, each on `threads_per_proof` host threads (1 = one pinned
        single-threaded proof per thread).  Returns (proofs or None, statuses, per-proof seconds, wall seconds)."""
        inp = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, self.num_inputs)
        n = inp.shape[0]
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert sd.shape == (n,)
        proofs = np.zeros((n, self.proof_words), dtype=np.uint64) if want_proofs else None
        st = np.full(n, -1, dtype=np.int32)
        per = np.zeros(n, dtype=np.float64)
        wall = self.lib.p25o_prove_many_grouped(self.h, _p(inp), _p(sd), n, int(threads), int(threads_per_proof),
                                                _p(proofs), _p(st), _p(per))
        return proofs, st, per, float(wall)

    def verify(self, proof, digest=None, cs_cap=None):
        if digest is None:
            digest, cs_cap = self.digest()
        msg = C.create_string_buffer(512)
        st = self.lib.p25o_verify(self.h, _p(np.ascontiguousarray(digest, dtype=np.uint64)),
                                  _p(np.ascontiguousarray(cs_cap, dtype=np.uint64)),
                                  _p(np.ascontiguousarray(proof, dtype=np.uint64)), msg, 512)
        return st, msg.value.decode()


class Oracle:
    """ctypes view of oracle/libp25_oracle.so -- the CPU restatement (checker only)."""

    def __init__(self):
        path = os.path.join(ORACLE_DIR, "libp25_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        self.lib = C.CDLL(path)
        L = self.lib
        vp, sz, ui, u64 = C.c_void_p, C.c_size_t, C.c_uint, C.c_uint64
        L.p25o_poseidon_permute.argtypes = [vp, sz]
        L.p25o_poseidon2_permute.argtypes = [vp, sz]
        L.p25o_poseidon2_trace.argtypes = [vp, vp]
        L.p25o_poseidon_trace.argtypes = [vp, vp]
        L.p25o_poseidon_fast_partial_inputs.argtypes = [vp, vp]
        L.p25o_hash_no_pad.argtypes = [vp, sz, vp]
        L.p25o_mul.argtypes = [u64, u64]
        L.p25o_mul.restype = u64
        L.p25o_inv.argtypes = [u64]
        L.p25o_inv.restype = u64
        L.p25o_merkle_commit.argtypes = [vp, sz, sz, ui, vp, vp]
        L.p25o_lde_commit.argtypes = [vp, ui, sz, C.c_int, ui, ui, vp, vp, vp]
        L.p25o_set_threads.argtypes = [C.c_int]
        L.p25o_circuit_load.restype = vp
        L.p25o_circuit_load.argtypes = [C.c_char_p, sz]
        L.p25o_circuit_free.argtypes = [vp]
        L.p25o_circuit_info.argtypes = [vp, vp]
        L.p25o_witness.argtypes = [vp, vp, u64, vp, C.c_char_p, sz]
        L.p25o_witness_forced.argtypes = [vp, vp, u64, vp, C.c_char_p, sz]
        L.p25o_check_constraints.argtypes = [vp, vp, C.c_char_p, sz]
        L.p25o_check_constraints.restype = C.c_long
        L.p25o_precompute.argtypes = [vp]
        L.p25o_circuit_digest.argtypes = [vp, vp, vp]
        L.p25o_proof_words.argtypes = [vp]
        L.p25o_proof_words.restype = sz
        L.p25o_prove.argtypes = [vp, vp, u64, vp, vp, C.c_char_p, sz]
        L.p25o_verify.argtypes = [vp, vp, vp, vp, C.c_char_p, sz]
        L.p25o_prove_many.argtypes = [vp, vp, vp, sz, C.c_int, vp, vp, vp]
        L.p25o_prove_many_grouped.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, vp, vp, vp]
        L.p25o_prove_many_grouped.restype = C.c_double
        L.p25o_stage_shapes.argtypes = [vp, vp, vp]
        L.p25o_eval_polys.argtypes = [vp, sz, sz, vp, u64, vp]
        L.p25o_eval_gate.argtypes = [ui, C.c_int, vp, vp, vp, C.c_int, vp]
        L.p25o_prove_filler.argtypes = [vp, vp, vp, vp, C.c_char_p, sz]
        L.p25o_num_random_fill.argtypes = [vp]
        L.p25o_num_random_fill.restype = sz
        L.p25o_transcript.argtypes = [vp, vp, vp, sz, vp]
        L.p25o_partial_products.argtypes = [vp, vp, vp, vp, vp]
        L.p25o_quotient.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.p25o_fri_prove.argtypes = [vp, ui, ui, ui, vp, sz, ui, ui, vp, sz, vp, sz]
        L.p25o_fri_prove.restype = sz
        L.p25o_prove_many.restype = C.c_double
        L.p25o_set_tuned.argtypes = [C.c_int]
        L.p25o_set_tuned.restype = C.c_int
        L.p25o_x8_available.restype = C.c_int
        L.p25o_poseidon_permute_x8.argtypes = [vp]
        L.p25o_set_threads(min(64, os.cpu_count() or 1))

    def set_threads(self, n):
        self.lib.p25o_set_threads(n)

    def set_tuned(self, on):
        """The tuned cpu_baseline leg: Merkle trees hashed eight at a time on AVX-512 (ref_hash_x8.cpp).  Process-wide;
        returns whether it is now in effect (False: switched off, or the CPU has no AVX-512).  The checker leaves it off."""
        return bool(self.lib.p25o_set_tuned(1 if on else 0))

    def x8_available(self):
        return bool(self.lib.p25o_x8_available())

    def poseidon_permute_x8(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).reshape(8, 12).copy()
        self.lib.p25o_poseidon_permute_x8(_p(s))
        return s

    def load_circuit(self, blob):
        return OracleCircuit(self.lib, blob)

    def eval_gate(self, kind, wires, consts, pih, base=False):
        """Constraints of one row of gate `kind`: wires [num_wires][2], consts [2][2], pih [4] -> [n][2]."""
        w = np.ascontiguousarray(wires, dtype=np.uint64)
        k = np.ascontiguousarray(consts, dtype=np.uint64)
        h = np.ascontiguousarray(pih, dtype=np.uint64)
        out = np.zeros((512, 2), dtype=np.uint64)
        n = self.lib.p25o_eval_gate(kind, w.shape[0], _p(w), _p(k), _p(h), int(base), _p(out))
        return out[:n].copy()

    def transcript(self, segments):
        """segments: [(words_to_observe, n_challenges), ...] -> all challenges drawn, in order."""
        obs = np.ascontiguousarray(np.concatenate([np.asarray(w, dtype=np.uint64).ravel() for w, _ in segments]
                                                  + [np.zeros(0, dtype=np.uint64)]))
        lens = np.array([np.asarray(w).size for w, _ in segments], dtype=np.uint32)
        nch = np.array([k for _, k in segments], dtype=np.uint32)
        out = np.zeros(int(nch.sum()), dtype=np.uint64)
        self.lib.p25o_transcript(_p(obs), _p(lens), _p(nch), len(segments), _p(out))
        return out

    def eval_polys(self, coeffs, point, scale=1):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64)
        pt = np.ascontiguousarray(point, dtype=np.uint64)
        out = np.zeros((a.shape[0], 2), dtype=np.uint64)
        self.lib.p25o_eval_polys(_p(a), a.shape[0], a.shape[1], _p(pt), int(scale), _p(out))
        return out

    def fri_prove(self, coeffs, rate_bits, cap_height, arity_bits, pow_bits, num_queries, seed):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64)
        assert a.ndim == 2 and a.shape[0] == 2
        log_n = int(a.shape[1]).bit_length() - 1
        ar = np.array(arity_bits, dtype=np.int32)
        sd = np.ascontiguousarray(seed, dtype=np.uint64)
        out = np.zeros(1 << 22, dtype=np.uint64)
        n = self.lib.p25o_fri_prove(_p(a), log_n, rate_bits, cap_height, _p(ar), len(ar), pow_bits, num_queries,
                                    _p(sd), sd.size, _p(out), out.size)
        assert n > 0, "oracle FRI failed"
        return out[:n].copy()

    def poseidon_permute(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).copy().reshape(-1, 12)
        self.lib.p25o_poseidon_permute(_p(s), s.shape[0])
        return s

    def poseidon2_permute(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).copy().reshape(-1, 12)
        self.lib.p25o_poseidon2_permute(_p(s), s.shape[0])
        return s

    def poseidon2_trace(self, state):
        s = np.ascontiguousarray(state, dtype=np.uint64).copy()
        tr = np.zeros(106, dtype=np.uint64)
        self.lib.p25o_poseidon2_trace(_p(s), _p(tr))
        return s, tr

    def poseidon_trace(self, state):
        """Poseidon v1, naive form: (output state, the 106 S-box inputs in PoseidonGate wire order)."""
        s = np.ascontiguousarray(state, dtype=np.uint64).copy()
        tr = np.zeros(106, dtype=np.uint64)
        self.lib.p25o_poseidon_trace(_p(s), _p(tr))
        return s, tr

    def poseidon_fast_partial_inputs(self, state):
        """Poseidon v1 in upstream's fast-partial-rounds form: (output, lane 0's 22 partial-round S-box inputs)."""
        s = np.ascontiguousarray(state, dtype=np.uint64).copy()
        pi = np.zeros(22, dtype=np.uint64)
        self.lib.p25o_poseidon_fast_partial_inputs(_p(s), _p(pi))
        return s, pi

    def hash_no_pad(self, words):
        a = np.ascontiguousarray(words, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.lib.p25o_hash_no_pad(_p(a), a.size, _p(out))
        return out

    def merkle_commit(self, leaves_rm, cap_height, want_tree=False):
        a = np.ascontiguousarray(leaves_rm, dtype=np.uint64)
        n, w = a.shape
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        words, m = 0, n
        while m >= (1 << cap_height):
            words += 4 * m
            if m == 1:
                break
            m >>= 1
        tree = np.zeros(words, dtype=np.uint64) if want_tree else None
        self.lib.p25o_merkle_commit(_p(a), n, w, cap_height, _p(cap), _p(tree))
        return (cap, tree) if want_tree else cap

    def lde_commit(self, polys, rate_bits, cap_height, from_coeffs=False):
        a = np.ascontiguousarray(polys, dtype=np.uint64)
        npolys, n = a.shape
        log_n = n.bit_length() - 1
        coeffs = np.zeros_like(a)
        lde = np.zeros((npolys, n << rate_bits), dtype=np.uint64)
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        self.lib.p25o_lde_commit(_p(a), log_n, npolys, int(from_coeffs), rate_bits, cap_height,
                                 _p(coeffs), _p(lde), _p(cap))
        return coeffs, lde, cap


def splitmix_field(n, seed=0x243F6A8885A308D3):
    """n canonical Goldilocks elements from SplitMix64 (SURVEY.md 8d synthetic-input recipe)."""
    out = np.empty(n, dtype=np.uint64)
    x = np.uint64(seed)
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out[:] = z % np.uint64(P)
    return out
