"""ctypes binding of libp25.so (C ABI: include/p25.h)."""
import ctypes as C
import os
import numpy as np

__all__ = ["P25Error", "lib", "lib_path", "device_init", "poseidon_permute", "poseidon2_permute",
           "merkle_commit", "merkle_tree_words", "lde_commit", "EXPORTED_SYMBOLS", "P"]

P = 0xFFFFFFFF00000001
_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "libp25.so")

STATUS_NAMES = {0: "OK", 1: "INVALID_ARG", 2: "NO_DEVICE", 3: "HIP", 4: "WITNESS_CONFLICT",
                5: "GENERATORS_NOT_RUN", 6: "OPENING_IN_SUBGROUP", 7: "INTERNAL", 8: "PARSE"}


class P25Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"p25 status {status} ({STATUS_NAMES.get(status, '?')}): {msg}")
        self.status = status


_lib = None
u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); every symbol include/p25.h declares
EXPORTED_SYMBOLS = {
    "p25_last_error": (C.c_char_p, []),
    "p25_version": (C.c_char_p, []),
    "p25_device_init": (C.c_int32, [C.c_int]),
    "p25_poseidon_permute": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "p25_poseidon2_permute": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "p25_merkle_commit": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p]),
    "p25_merkle_tree_words": (C.c_size_t, [C.c_size_t, C.c_uint]),
    "p25_lde_commit": (C.c_int32, [C.c_void_p, C.c_uint, C.c_size_t, C.c_int, C.c_uint, C.c_uint,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "p25_merkle_commit_dev": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint,
                                          C.c_void_p, C.c_void_p]),
    "p25_lde_commit_dev": (C.c_int32, [C.c_void_p, C.c_uint, C.c_size_t, C.c_int, C.c_uint, C.c_uint,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "p25_poseidon_permute_dev": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_void_p]),
}


def lib():
    """Load libp25.so (fails loudly if it was not built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise P25Error(7, f"{lib_path} not found -- run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(lib_path)
        for name, (res, args) in EXPORTED_SYMBOLS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _check(status):
    if status != 0:
        raise P25Error(status, lib().p25_last_error().decode())


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_init(index=0):
    _check(lib().p25_device_init(index))


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
