"""ctypes binding of libp25.so (C ABI: include/p25.h)."""
import ctypes as C
import os

import numpy as np

__all__ = ["P25Error", "lib", "lib_path", "device_init", "shader_clock_hz", "poseidon_permute", "poseidon2_permute",
           "merkle_commit", "merkle_tree_words", "lde_commit", "EXPORTED_SYMBOLS", "P",
           "P3Config", "Circuit", "p3_proof_from_json", "Timings", "p3_prove_fibonacci", "p3_inputs_to_json",
           "Air", "p3_prove_air", "transcript", "fri_prove", "eval_polys", "RuntimeInfo", "runtime_info", "Comm",
           "comm_unique_id", "WARN_HW_QUEUES_LATE"]

P = 0xFFFFFFFF00000001
_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, in-tree, and nothing else: no environment override, no fallback.  (Profiling tools that need
# another build -- tools/qmask.sh's gate-mask library -- assign `binding.lib_path` explicitly before the first call.)
lib_path = os.path.join(_HERE, "libp25.so")

STATUS_NAMES = {0: "OK", 1: "INVALID_ARG", 2: "NO_DEVICE", 3: "HIP", 4: "WITNESS_CONFLICT",
                5: "GENERATORS_NOT_RUN", 6: "OPENING_IN_SUBGROUP", 7: "INTERNAL", 8: "PARSE",
                9: "WARN_HW_QUEUES_LATE", 10: "RCCL"}
WARN_HW_QUEUES_LATE = 9
COMM_ID_BYTES = 128


class P25Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"p25 status {status} ({STATUS_NAMES.get(status, '?')}): {msg}")
        self.status = status


class P3Config(C.Structure):
    """p25_p3_config: FriConfig (src/p3/mod.rs:242-246) + P3Config (src/p3/mod.rs:74-87)."""
    _fields_ = [(n, C.c_int32) for n in ("log_blowup", "num_queries", "proof_of_work_bits",
                                         "log_quotient_degree", "log_trace_height", "trace_width",
                                         "opening_matrix_log_max_height", "quotient_opened_len", "degree_bits")]

    @classmethod
    def fib64(cls):
        return cls(1, 100, 16, 0, 6, 3, 7, 2, 6)


class AirNode(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("reserved", C.c_uint32), ("value", C.c_uint64)]


class AirConstraint(C.Structure):
    _fields_ = [("node", C.c_uint32), ("when", C.c_uint32)]


class AirC(C.Structure):
    _fields_ = [("width", C.c_uint32), ("n_nodes", C.c_uint32), ("n_constraints", C.c_uint32), ("reserved", C.c_uint32),
                ("nodes", C.POINTER(AirNode)), ("constraints", C.POINTER(AirConstraint))]


class Air:
    """An AIR as an expression DAG (p25_air in include/p25.h): the data form of the reference's `Air` trait
    (src/p3/air.rs:10-18).  Build it like the trait's `eval` body:

        air = Air(3); a, b, c = air.local(0), air.local(1), air.local(2)
        air.assert_zero(air.sub(air.add(a, b), c)); air.when_first_row(air.sub(air.const(1), a)); ...
    """
    ALWAYS, FIRST_ROW, LAST_ROW, TRANSITION = 0, 1, 2, 3

    def __init__(self, width):
        self.width, self.nodes, self.constraints = width, [], []

    def _n(self, op, a=0, b=0, value=0):
        self.nodes.append((op, a, b, value))
        return len(self.nodes) - 1

    def local(self, col): return self._n(0, col)
    def next(self, col): return self._n(1, col)
    def const(self, v): return self._n(2, 0, 0, int(v))
    def add(self, x, y): return self._n(3, x, y)
    def sub(self, x, y): return self._n(4, x, y)
    def mul(self, x, y): return self._n(5, x, y)
    def assert_zero(self, x, when=0): self.constraints.append((x, when))
    def when_first_row(self, x): self.assert_zero(x, 1)
    def when_last_row(self, x): self.assert_zero(x, 2)
    def when_transition(self, x): self.assert_zero(x, 3)

    @classmethod
    def fibonacci(cls):
        """src/p3/mod.rs:176-221, constraint for constraint."""
        air = cls(3)
        la, lb, lc = air.local(0), air.local(1), air.local(2)
        na, nb = air.next(0), air.next(1)
        air.assert_zero(air.sub(air.add(la, lb), lc))
        one = air.const(1)
        air.when_first_row(air.sub(one, la))
        air.when_first_row(air.sub(one, lb))
        air.when_transition(air.sub(na, lb))
        air.when_transition(air.sub(nb, lc))
        return air

    def to_c(self):
        nodes = (AirNode * len(self.nodes))(*[AirNode(op, a, b, 0, v) for op, a, b, v in self.nodes])
        cons = (AirConstraint * len(self.constraints))(*[AirConstraint(n, w) for n, w in self.constraints])
        c = AirC(self.width, len(self.nodes), len(self.constraints), 0, nodes, cons)
        c._keep = (nodes, cons)
        return c


class CircuitInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("degree_bits", "num_rows_used", "num_wires", "num_routed_wires",
                                          "num_inputs", "num_generators", "num_gate_types", "num_selectors",
                                          "num_constants_sigmas", "num_gate_constraints", "proof_words",
                                          "witness_levels", "witness_slots", "num_random_fill",
                                          "num_challenges", "num_partial_products", "quotient_degree_factor",
                                          "num_public_inputs")]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("witness_ms", "wires_commit_ms", "partial_products_ms", "zs_commit_ms",
                                         "quotient_ms", "quotient_commit_ms", "openings_ms", "fri_ms", "total_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


class RuntimeInfo(C.Structure):
    """p25_runtime_info_t: what the process actually runs with (hardware-queue request, stream pool)."""
    _fields_ = [(n, C.c_int32) for n in ("device_index", "hw_queues_requested", "hw_queues_env", "hw_queues_host_exported",
                                         "runtime_open_before_init", "hw_queues_setting_late", "proving_streams",
                                         "main_streams")] + [("reserved", C.c_int32 * 8)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


_lib = None
vp, sz, ui, i32 = C.c_void_p, C.c_size_t, C.c_uint, C.c_int32

# name -> (restype, argtypes); every symbol include/p25.h declares
EXPORTED_SYMBOLS = {
    "p25_last_error": (C.c_char_p, []),
    "p25_shader_clock_hz": (i32, [C.POINTER(C.c_double)]),
    "p25_circuit_set_streams": (i32, [vp, i32]),
    "p25_circuit_build_aggregator": (i32, [vp, vp, vp, i32, C.POINTER(vp)]),
    "p25_circuit_to_bytes": (i32, [vp, C.POINTER(vp), C.POINTER(sz)]),
    "p25_circuit_from_bytes": (i32, [vp, sz, vp, sz, vp, C.POINTER(vp)]),
    "p25_circuit_input_targets": (i32, [vp, vp, sz, C.POINTER(sz)]),
    "p25_free": (None, [vp]),
    "p25_version": (C.c_char_p, []),
    "p25_device_init": (i32, [C.c_int]),
    "p25_poseidon_permute": (i32, [vp, sz]),
    "p25_poseidon2_permute": (i32, [vp, sz]),
    "p25_merkle_commit": (i32, [vp, sz, sz, ui, vp, vp]),
    "p25_merkle_tree_words": (sz, [sz, ui]),
    "p25_lde_commit": (i32, [vp, ui, sz, C.c_int, ui, ui, vp, vp, vp]),
    "p25_merkle_commit_dev": (i32, [vp, sz, sz, sz, ui, vp, vp]),
    "p25_lde_commit_dev": (i32, [vp, ui, sz, C.c_int, ui, ui, vp, vp, vp, vp, vp]),
    "p25_poseidon_permute_dev": (i32, [vp, sz, vp]),
    "p25_circuit_build_p3_verifier": (i32, [C.POINTER(P3Config), i32, C.POINTER(vp)]),
    "p25_circuit_build_gadget": (i32, [i32, i32, C.POINTER(vp)]),
    "p25_circuit_build_gate_eval": (i32, [i32, C.POINTER(vp)]),
    "p25_circuit_build_recursive_verifier": (i32, [vp, vp, vp, i32, C.POINTER(vp)]),
    "p25_circuit_export": (i32, [vp, vp, sz, C.POINTER(sz)]),
    "p25_circuit_import": (i32, [vp, sz, C.POINTER(vp)]),
    "p25_circuit_destroy": (None, [vp]),
    "p25_circuit_info": (i32, [vp, C.POINTER(CircuitInfo)]),
    "p25_circuit_gate_counts": (i32, [vp, vp, sz, C.c_char_p, sz]),
    "p25_circuit_digest": (i32, [vp, vp, vp]),
    "p25_prove_batch": (i32, [vp, vp, sz, vp, vp, sz, vp, C.POINTER(Timings)]),
    "p25_prove_batch_filler": (i32, [vp, vp, sz, vp, vp, sz, vp]),
    "p25_prove_batch_dev": (i32, [vp, vp, sz, vp, vp, sz, vp, C.POINTER(Timings)]),
    "p25_prove_batch_dev_windows": (i32, [vp, vp, sz, sz, sz, vp, vp, sz, vp]),
    "p25_device_init_ex": (i32, [C.c_int, C.c_int]),
    "p25_circuit_sync": (i32, [vp]),
    "p25_circuit_stream_join": (i32, [vp, vp]),
    "p25_circuit_wait_stream": (i32, [vp, vp]),
    "p25_circuit_mark": (i32, [vp, C.c_uint32]),
    "p25_circuit_wait_mark": (i32, [vp, vp, C.c_uint32]),
    "p25_circuit_stream_wait_mark": (i32, [vp, C.c_uint32, vp]),
    "p25_circuit_kernel_stats": (i32, [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "p25_witness": (i32, [vp, vp, C.c_uint64, vp, C.POINTER(i32)]),
    "p25_transcript": (i32, [vp, vp, vp, sz, vp]),
    "p25_partial_products": (i32, [vp, vp, vp, vp, vp]),
    "p25_quotient": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    "p25_eval_polys": (i32, [vp, sz, ui, vp, C.c_uint64, vp]),
    "p25_fri_prove_words": (sz, [ui, ui, ui, vp, sz, ui]),
    "p25_fri_prove": (i32, [vp, ui, ui, ui, vp, sz, ui, ui, vp, sz, vp, sz, C.POINTER(i32)]),
    "p25_p3_proof_from_json": (i32, [C.c_char_p, sz, vp, sz, C.POINTER(sz), C.POINTER(P3Config)]),
    "p25_proof_to_json": (i32, [vp, vp, vp, sz, C.POINTER(sz)]),
    "p25_proof_to_bytes": (i32, [vp, vp, vp, sz, C.POINTER(sz)]),
    "p25_proof_from_bytes": (i32, [vp, vp, sz, vp, sz]),
    "p25_p3_prove_fibonacci": (i32, [i32, i32, i32, C.c_uint64, i32, vp, sz, C.POINTER(sz), C.POINTER(P3Config)]),
    "p25_p3_inputs_to_json": (i32, [vp, sz, C.POINTER(P3Config), vp, sz, C.POINTER(sz)]),
    "p25_circuit_build_p3_verifier_air": (i32, [C.POINTER(P3Config), C.POINTER(AirC), C.POINTER(vp)]),
    "p25_p3_prove_air_ex": (i32, [C.POINTER(AirC), vp, i32, i32, i32, i32, C.c_uint64, i32, vp, sz, C.POINTER(sz),
                                  C.POINTER(P3Config)]),
    "p25_runtime_info": (i32, [C.POINTER(RuntimeInfo)]),
    "p25_comm_unique_id": (i32, [vp]),
    "p25_comm_init": (i32, [vp, i32, i32, C.POINTER(vp)]),
    "p25_comm_destroy": (i32, [vp]),
    "p25_comm_rank": (i32, [vp]),
    "p25_comm_world": (i32, [vp]),
    "p25_comm_stream": (vp, [vp]),
    "p25_comm_sync": (i32, [vp]),
    "p25_comm_barrier": (i32, [vp]),
    "p25_comm_max_f64": (i32, [vp, C.POINTER(C.c_double)]),
    "p25_gather_proofs": (i32, [vp, vp, i32, vp, sz, vp, C.POINTER(sz), i32, vp, vp]),
    "p25_p3_prove_air": (i32, [C.POINTER(AirC), vp, i32, i32, i32, C.c_uint64, i32, vp, sz, C.POINTER(sz),
                               C.POINTER(P3Config)]),
}


def lib():
    """Load libp25.so (fails loudly if it was not built: there is no fallback)."""
    global _lib
    if _lib is None:
        path = globals()["lib_path"]
        if not os.path.exists(path):
            raise P25Error(7, f"{path} not found -- run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(path)
        for name, (res, args) in EXPORTED_SYMBOLS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _check(status):
    if status != 0:
        raise P25Error(status, lib().p25_last_error().decode())


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_init(index=0, hw_queues=None):
    """p25_device_init / p25_device_init_ex: hw_queues None = the library's default (24), 0 = leave GPU_MAX_HW_QUEUES alone.
    Returns True when the hardware-queue request probably came too late to have an effect (P25_WARN_HW_QUEUES_LATE: the process
    had the GPU runtime open already and GPU_MAX_HW_QUEUES was not exported) -- also readable from runtime_info()."""
    if hw_queues is None:
        _check(lib().p25_device_init(index))
        return bool(runtime_info().hw_queues_setting_late)
    st = lib().p25_device_init_ex(index, hw_queues)
    if st == WARN_HW_QUEUES_LATE:
        return True
    _check(st)
    return False


def runtime_info():
    ri = RuntimeInfo()
    _check(lib().p25_runtime_info(C.byref(ri)))
    return ri


def comm_unique_id():
    """p25_comm_unique_id: 128 bytes rank 0 hands to every rank out of band (ncclGetUniqueId)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().p25_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """p25_comm: the library-owned RCCL communicator + side stream of the final aggregation step (one rank per GPU)."""

    def __init__(self, unique_id, rank, world):
        assert len(unique_id) == COMM_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().p25_comm_init(buf, rank, world, C.byref(h)))
        self.h, self.rank, self.world = h, rank, world

    @property
    def stream(self):
        return lib().p25_comm_stream(self.h)

    def gather(self, circuit, mark_slot, d_proofs, proof_stride, d_status, counts, dst, d_all_proofs, d_all_status):
        """p25_gather_proofs with device addresses (ints); circuit may be None (no device-side wait)."""
        cnt = (C.c_size_t * self.world)(*[int(x) for x in counts])
        _check(lib().p25_gather_proofs(self.h, circuit._h if circuit is not None else None, int(mark_slot), d_proofs,
                                       proof_stride, d_status, cnt, dst, d_all_proofs, d_all_status))

    def sync(self):
        _check(lib().p25_comm_sync(self.h))

    def barrier(self):
        _check(lib().p25_comm_barrier(self.h))

    def max_f64(self, value):
        v = C.c_double(float(value))
        _check(lib().p25_comm_max_f64(self.h, C.byref(v)))
        return float(v.value)

    def close(self):
        if self.h:
            h, self.h = self.h, None
            _check(lib().p25_comm_destroy(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shader_clock_hz():
    hz = C.c_double(0)
    _check(lib().p25_shader_clock_hz(C.byref(hz)))
    return hz.value


def poseidon_permute(states):
    s = _u64(states).copy().reshape(-1, 12)
    _check(lib().p25_poseidon_permute(_ptr(s), s.shape[0]))
    return s


def poseidon2_permute(states):
    s = _u64(states).copy().reshape(-1, 12)
    _check(lib().p25_poseidon2_permute(_ptr(s), s.shape[0]))
    return s


def merkle_tree_words(n_leaves, cap_height):
    return lib().p25_merkle_tree_words(n_leaves, cap_height)


def merkle_commit(leaves_cm, cap_height, want_tree=False):
    """leaves_cm: [width][n_leaves] uint64 (column-major leaves).  Returns cap [2^cap][4] (and tree)."""
    a = _u64(leaves_cm)
    width, n = a.shape
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    tree = np.zeros(merkle_tree_words(n, cap_height), dtype=np.uint64) if want_tree else None
    _check(lib().p25_merkle_commit(_ptr(a), n, width, cap_height, _ptr(cap), _ptr(tree)))
    return (cap, tree) if want_tree else cap


def lde_commit(polys, rate_bits, cap_height, from_coeffs=False, want_lde=True):
    """polys: [n_polys][2^log_n].  Returns (coeffs, lde_bitrev, cap)."""
    a = _u64(polys)
    n_polys, n = a.shape
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    coeffs = np.zeros_like(a)
    lde = np.zeros((n_polys, n << rate_bits), dtype=np.uint64) if want_lde else None
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    _check(lib().p25_lde_commit(_ptr(a), log_n, n_polys, int(from_coeffs), rate_bits, cap_height,
                                _ptr(coeffs), _ptr(lde), _ptr(cap)))
    return coeffs, lde, cap


def transcript(segments):
    """Challenger script: segments = [(words_to_observe, n_challenges), ...] -> all challenges drawn, in order
    (upstream iop/challenger.rs)."""
    obs = np.ascontiguousarray(np.concatenate([np.asarray(w, dtype=np.uint64).ravel() for w, _ in segments]
                                              + [np.zeros(0, dtype=np.uint64)]))
    lens = np.array([np.asarray(w).size for w, _ in segments], dtype=np.uint32)
    nch = np.array([k for _, k in segments], dtype=np.uint32)
    out = np.zeros(int(nch.sum()), dtype=np.uint64)
    _check(lib().p25_transcript(_ptr(obs), _ptr(lens), _ptr(nch), len(segments), _ptr(out)))
    return out


def eval_polys(coeffs, point, scale=1):
    """Openings: coeffs[n_polys][2^log_n] evaluated at the extension point point*scale -> [n_polys][2]."""
    a = _u64(coeffs)
    n_polys, n = a.shape
    pt = _u64(point)
    out = np.zeros((n_polys, 2), dtype=np.uint64)
    _check(lib().p25_eval_polys(_ptr(a), n_polys, int(n).bit_length() - 1, _ptr(pt), int(scale), _ptr(out)))
    return out


def fri_prove(coeffs, rate_bits, cap_height, arity_bits, pow_bits, num_queries, seed):
    """FRI on one batched polynomial, coeffs[2][2^log_n] (extension components).  Returns (flat output, status);
    layout in include/p25.h (p25_fri_prove)."""
    a = _u64(coeffs)
    assert a.ndim == 2 and a.shape[0] == 2
    log_n = int(a.shape[1]).bit_length() - 1
    ar = np.array(arity_bits, dtype=np.int32)
    sd = _u64(seed)
    n = lib().p25_fri_prove_words(log_n, rate_bits, cap_height, _ptr(ar), len(ar), num_queries)
    if n == 0:
        raise P25Error(1, "bad FRI shape")
    out = np.zeros(n, dtype=np.uint64)
    st = i32(0)
    _check(lib().p25_fri_prove(_ptr(a), log_n, rate_bits, cap_height, _ptr(ar), len(ar), pow_bits, num_queries,
                               _ptr(sd), sd.size, _ptr(out), out.size, C.byref(st)))
    return out, st.value


def p3_proof_from_json(text):
    """plonky3 proof JSON (str/bytes) -> (inputs uint64[n], P3Config).  Mirrors
    serde_json::from_str::<P3ProofField> + Proof::set_witness (src/p3/mod.rs:233-234, 254-257)."""
    if isinstance(text, str):
        text = text.encode()
    n = sz(0)
    cfg = P3Config()
    _check(lib().p25_p3_proof_from_json(text, len(text), None, 0, C.byref(n), C.byref(cfg)))
    out = np.zeros(n.value, dtype=np.uint64)
    _check(lib().p25_p3_proof_from_json(text, len(text), _ptr(out), out.size, C.byref(n), C.byref(cfg)))
    return out, cfg


def p3_prove_fibonacci(log_n=6, num_queries=100, pow_bits=16, pow_start=0, threads=None):
    """Native plonky3 proof of the Fibonacci AIR with 2^log_n rows -> (inputs uint64[n], P3Config).
    With the defaults it reproduces the reference's artifacts/proof_fibonacci.json bit for bit."""
    threads = threads or min(16, os.cpu_count() or 1)
    n = sz(0)
    cfg = P3Config()
    _check(lib().p25_p3_prove_fibonacci(log_n, num_queries, pow_bits, pow_start, threads, None, 0, C.byref(n), C.byref(cfg)))
    out = np.zeros(n.value, dtype=np.uint64)
    _check(lib().p25_p3_prove_fibonacci(log_n, num_queries, pow_bits, pow_start, threads, _ptr(out), out.size,
                                        C.byref(n), C.byref(cfg)))
    return out, cfg


def p3_prove_air(air, trace, num_queries=100, pow_bits=16, pow_start=0, threads=None, log_blowup=1):
    """Native plonky3 proof of `air` (binding.Air) on `trace` (uint64[2^log_n][width]) -> (inputs, P3Config).
    log_blowup: FriConfig.log_blowup (1 = the reference's; 2 / 3 for AIRs of constraint degree up to 5 / 9)."""
    threads = threads or min(16, os.cpu_count() or 1)
    t = np.ascontiguousarray(trace, dtype=np.uint64)
    if t.ndim != 2 or t.shape[1] != air.width or t.shape[0] & (t.shape[0] - 1) or t.shape[0] < 2:
        raise ValueError("trace must be [2^log_n][air.width]")
    log_n = int(t.shape[0]).bit_length() - 1
    ac = air.to_c()
    n = sz(0)
    cfg = P3Config()
    _check(lib().p25_p3_prove_air_ex(C.byref(ac), None, log_n, log_blowup, num_queries, pow_bits, pow_start, threads, None, 0,
                                     C.byref(n), C.byref(cfg)))
    out = np.zeros(n.value, dtype=np.uint64)
    _check(lib().p25_p3_prove_air_ex(C.byref(ac), _ptr(t), log_n, log_blowup, num_queries, pow_bits, pow_start, threads, _ptr(out),
                                     out.size, C.byref(n), C.byref(cfg)))
    return out, cfg


def p3_inputs_to_json(inputs, cfg):
    a = _u64(inputs)
    n = sz(0)
    _check(lib().p25_p3_inputs_to_json(_ptr(a), a.size, C.byref(cfg), None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value)
    _check(lib().p25_p3_inputs_to_json(_ptr(a), a.size, C.byref(cfg), buf, n.value, C.byref(n)))
    return buf.raw[:n.value].decode()


class Circuit:
    """Built plonky2 circuit (upstream CircuitData); mirrors the reference's
    `builder.p3_verify_proof(..); let data = builder.build::<C>(); data.prove(pw)` (src/p3/mod.rs:239-260)."""

    def __init__(self, handle):
        self._h = vp(handle)
        self._info = None

    @classmethod
    def build_p3_verifier(cls, cfg=None, air=0):
        cfg = cfg or P3Config.fib64()
        h = vp()
        _check(lib().p25_circuit_build_p3_verifier(C.byref(cfg), air, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_p3_verifier_air(cls, cfg, air):
        """`builder.p3_verify_proof::<H>(proof, &air, fri_config)` for a user AIR (binding.Air)."""
        h = vp()
        ac = air.to_c()
        _check(lib().p25_circuit_build_p3_verifier_air(C.byref(cfg), C.byref(ac), C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_gadget(cls, kind, param=0):
        """kind: 0 and, 1 xor, 2 lsh, 3 rsh, 4 reverse_bits_len, 5 compress, 6 exp, 7 hash_iter_slices,
        8 two connected inputs (include/p25.h)."""
        h = vp()
        _check(lib().p25_circuit_build_gadget(kind, param, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_gate_eval(cls, kind):
        """Gate-level test circuit: evaluates gate `kind` in-circuit (eval_unfiltered_circuit) against expectations."""
        h = vp()
        _check(lib().p25_circuit_build_gate_eval(kind, C.byref(h)))
        return cls(h.value)

    def build_recursive_verifier(self, n_proofs=1, digest=None, cs_cap=None):
        """A circuit verifying `n_proofs` proofs of this circuit (upstream builder.verify_proof).  digest / cs_cap:
        this circuit's verifier data (default: computed on the GPU)."""
        h = vp()
        d = _u64(digest) if digest is not None else None
        cap = _u64(cs_cap) if cs_cap is not None else None
        _check(lib().p25_circuit_build_recursive_verifier(self._h, _ptr(d), _ptr(cap), n_proofs, C.byref(h)))
        return Circuit(h.value)

    def build_aggregator(self, n_proofs=2, digest=None, cs_cap=None):
        """build_recursive_verifier whose circuit registers 4 public inputs committing to the proofs it verifies."""
        h = vp()
        d = _u64(digest) if digest is not None else None
        cap = _u64(cs_cap) if cs_cap is not None else None
        _check(lib().p25_circuit_build_aggregator(self._h, _ptr(d), _ptr(cap), n_proofs, C.byref(h)))
        return Circuit(h.value)

    def public_inputs(self, proof):
        """The public inputs of a flat proof (its last num_public_inputs words)."""
        n = int(self.info.num_public_inputs)
        p = _u64(proof)
        return p[p.size - n:].copy() if n else np.zeros(0, dtype=np.uint64)

    @classmethod
    def from_blob(cls, blob):
        h = vp()
        buf = np.frombuffer(blob, dtype=np.uint8)
        _check(lib().p25_circuit_import(_ptr(buf), buf.size, C.byref(h)))
        return cls(h.value)

    def to_bytes(self):
        """upstream CircuitData::to_bytes (needs the GPU: the constants/sigmas commitment is part of the data)."""
        p, n = vp(), sz(0)
        _check(lib().p25_circuit_to_bytes(self._h, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value)
        finally:
            lib().p25_free(p)

    def input_target_indices(self):
        n = sz(0)
        _check(lib().p25_circuit_input_targets(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        _check(lib().p25_circuit_input_targets(self._h, _ptr(out), out.size, C.byref(n)))
        return out

    @classmethod
    def from_bytes(cls, data, input_targets):
        """upstream CircuitData::from_bytes + the per-proof input targets.  Returns (circuit, stored circuit digest)."""
        h = vp()
        buf = np.frombuffer(data, dtype=np.uint8)
        it = np.ascontiguousarray(input_targets, dtype=np.uint32)
        dg = np.zeros(4, dtype=np.uint64)
        _check(lib().p25_circuit_from_bytes(_ptr(buf), buf.size, _ptr(it), it.size, _ptr(dg), C.byref(h)))
        return cls(h.value), dg

    def to_blob(self):
        n = sz(0)
        _check(lib().p25_circuit_export(self._h, None, 0, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        _check(lib().p25_circuit_export(self._h, _ptr(buf), buf.size, C.byref(n)))
        return buf.tobytes()

    def close(self):
        if self._h:
            lib().p25_circuit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self):
        if self._info is None:
            ci = CircuitInfo()
            _check(lib().p25_circuit_info(self._h, C.byref(ci)))
            self._info = ci
        return self._info

    def gate_counts(self):
        counts = np.zeros(16, dtype=np.uint64)
        ids = C.create_string_buffer(4096)
        _check(lib().p25_circuit_gate_counts(self._h, _ptr(counts), 16, ids, 4096))
        names = [s for s in ids.value.decode().split("\n") if s]
        return {n: int(c) for n, c in zip(names, counts)}

    def digest(self):
        d = np.zeros(4, dtype=np.uint64)
        cap = np.zeros((16, 4), dtype=np.uint64)
        _check(lib().p25_circuit_digest(self._h, _ptr(d), _ptr(cap)))
        return d, cap

    def witness(self, inputs, seed=0):
        inp = _u64(inputs)
        n = 1 << int(self.info.degree_bits)
        wires = np.zeros((int(self.info.num_wires), n), dtype=np.uint64)
        st = i32(0)
        _check(lib().p25_witness(self._h, _ptr(inp), seed, _ptr(wires), C.byref(st)))
        return wires, st.value

    def partial_products(self, wires, betas, gammas):
        """Z and partial products from witness values: wires [num_wires][n] -> [NC*(1+NP)][n]."""
        w, b, g = _u64(wires), _u64(betas), _u64(gammas)
        n = 1 << int(self.info.degree_bits)
        assert w.shape == (int(self.info.num_wires), n) and b.size == 2 and g.size == 2
        rows = int(self.info.num_challenges) * (1 + int(self.info.num_partial_products))   # what the C side writes
        out = np.zeros((rows, n), dtype=np.uint64)
        _check(lib().p25_partial_products(self._h, _ptr(w), _ptr(b), _ptr(g), _ptr(out)))
        return out

    def quotient(self, wires, zs_pp, betas, gammas, alphas):
        """Quotient chunk coefficients [NC*8][n] from witness and Z/partial-product values."""
        w, z, b, g, a = _u64(wires), _u64(zs_pp), _u64(betas), _u64(gammas), _u64(alphas)
        n = 1 << int(self.info.degree_bits)
        nc, npp, qdf = (int(self.info.num_challenges), int(self.info.num_partial_products),
                        int(self.info.quotient_degree_factor))
        assert w.shape == (int(self.info.num_wires), n) and z.shape == (nc * (1 + npp), n)
        out = np.zeros((nc * qdf, n), dtype=np.uint64)
        _check(lib().p25_quotient(self._h, _ptr(w), _ptr(z), _ptr(b), _ptr(g), _ptr(a), _ptr(out)))
        return out

    def prove(self, inputs, seeds=None, timings=False):
        """inputs: [n_proofs][num_inputs].  Returns (proofs [n_proofs][proof_words], statuses[, timings])."""
        inp = _u64(inputs)
        if inp.ndim == 1:
            inp = inp.reshape(1, -1)
        n = inp.shape[0]
        assert inp.shape[1] == int(self.info.num_inputs)
        pw = int(self.info.proof_words)
        proofs = np.zeros((n, pw), dtype=np.uint64)
        st = np.zeros(n, dtype=np.int32)
        sd = _u64(seeds) if seeds is not None else None
        tm = Timings()
        _check(lib().p25_prove_batch(self._h, _ptr(inp), n, _ptr(sd), _ptr(proofs), pw, _ptr(st),
                                     C.byref(tm) if timings else None))
        return (proofs, st, tm) if timings else (proofs, st)

    def prove_filler(self, inputs, filler):
        """prove() with explicit RandomValueGenerator values filler[n_proofs][num_random_fill] instead of seeds."""
        inp = _u64(inputs).reshape(-1, int(self.info.num_inputs))
        f = _u64(filler).reshape(inp.shape[0], int(self.info.num_random_fill))
        pw = int(self.info.proof_words)
        proofs = np.zeros((inp.shape[0], pw), dtype=np.uint64)
        st = np.zeros(inp.shape[0], dtype=np.int32)
        _check(lib().p25_prove_batch_filler(self._h, _ptr(inp), inp.shape[0], _ptr(f), _ptr(proofs), pw, _ptr(st)))
        return proofs, st

    def prove_dev(self, d_inputs, n_proofs, d_seeds, d_proofs, proof_stride, d_status, timings=None):
        """Device-resident batch (raw device addresses, e.g. torch tensor .data_ptr())."""
        _check(lib().p25_prove_batch_dev(self._h, d_inputs, n_proofs, d_seeds, d_proofs, proof_stride, d_status,
                                         C.byref(timings) if timings is not None else None))

    def prove_dev_windows(self, d_buffer, window_stride, last_window_offset, n_proofs, d_seeds, d_proofs, proof_stride, d_status):
        """p25_prove_batch_dev_windows: proof i reads its inputs at d_buffer + min(i * window_stride, last_window_offset) words."""
        _check(lib().p25_prove_batch_dev_windows(self._h, d_buffer, window_stride, last_window_offset, n_proofs, d_seeds,
                                                 d_proofs, proof_stride, d_status))

    def sync(self):
        _check(lib().p25_circuit_sync(self._h))

    def stream_join(self, stream):
        """`stream` (a raw hipStream_t, e.g. torch.cuda.Stream.cuda_stream) waits for every proof enqueued so far."""
        _check(lib().p25_circuit_stream_join(self._h, C.c_void_p(stream)))

    def wait_stream(self, stream):
        """Proofs requested from now on start only after what `stream` holds now."""
        _check(lib().p25_circuit_wait_stream(self._h, C.c_void_p(stream)))

    def mark(self, slot):
        """Remember the tail of every proving stream under `slot` (0..15, P25_MAX_MARKS); see wait_mark."""
        _check(lib().p25_circuit_mark(self._h, slot))

    def stream_wait_mark(self, slot, stream):
        """`stream` (a raw hipStream_t) waits for mark(slot)'s point."""
        _check(lib().p25_circuit_stream_wait_mark(self._h, slot, C.c_void_p(stream)))

    def wait_mark(self, producer, slot):
        """Proofs requested from now on start only after `producer.mark(slot)`'s point (device-side, events only)."""
        _check(lib().p25_circuit_wait_mark(self._h, producer._h, slot))

    def set_streams(self, n):
        """Proofs kept in flight by the batch entry points (1..32, default 16)."""
        _check(lib().p25_circuit_set_streams(self._h, n))

    def kernel_stats(self, enable=True, reset=False):
        """(ms, launches) of the dominant kernel (wires leaf sponge), measured with HIP events."""
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(lib().p25_circuit_kernel_stats(self._h, int(enable), int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def proof_to_bytes(self, proof):
        """upstream ProofWithPublicInputs::to_bytes() of a flat proof."""
        p = _u64(proof)
        n = sz(0)
        _check(lib().p25_proof_to_bytes(self._h, _ptr(p), None, 0, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        _check(lib().p25_proof_to_bytes(self._h, _ptr(p), _ptr(buf), buf.size, C.byref(n)))
        return buf.tobytes()

    def proof_from_bytes(self, data):
        buf = np.frombuffer(data, dtype=np.uint8)
        out = np.zeros(int(self.info.proof_words), dtype=np.uint64)
        _check(lib().p25_proof_from_bytes(self._h, _ptr(buf), buf.size, _ptr(out), out.size))
        return out

    def proof_to_json(self, proof):
        p = _u64(proof)
        n = sz(0)
        _check(lib().p25_proof_to_json(self._h, _ptr(p), None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(lib().p25_proof_to_json(self._h, _ptr(p), buf, n.value, C.byref(n)))
        return buf.raw[:n.value].decode()
