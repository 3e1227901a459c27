import ctypes as C
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_p25():
    """Import the package directory `plonky2.5_amd/` (dot in the name) as module `plonky25_amd`."""
    name = "plonky25_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "plonky2.5_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Oracle:
    """ctypes view of oracle/libp25_oracle.so -- the CPU restatement (checker only)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "libp25_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.lib = C.CDLL(path)
        L = self.lib
        vp, sz, ui = C.c_void_p, C.c_size_t, C.c_uint
        L.p25o_poseidon_permute.argtypes = [vp, sz]
        L.p25o_poseidon2_permute.argtypes = [vp, sz]
        L.p25o_poseidon2_trace.argtypes = [vp, vp]
        L.p25o_hash_no_pad.argtypes = [vp, sz, vp]
        L.p25o_mul.argtypes = [C.c_uint64, C.c_uint64]
        L.p25o_mul.restype = C.c_uint64
        L.p25o_inv.argtypes = [C.c_uint64]
        L.p25o_inv.restype = C.c_uint64
        L.p25o_merkle_commit.argtypes = [vp, sz, sz, ui, vp, vp]
        L.p25o_lde_commit.argtypes = [vp, ui, sz, C.c_int, ui, ui, vp, vp, vp]

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def poseidon_permute(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).copy().reshape(-1, 12)
        self.lib.p25o_poseidon_permute(self._p(s), s.shape[0])
        return s

    def poseidon2_permute(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).copy().reshape(-1, 12)
        self.lib.p25o_poseidon2_permute(self._p(s), s.shape[0])
        return s

    def poseidon2_trace(self, state):
        s = np.ascontiguousarray(state, dtype=np.uint64).copy()
        tr = np.zeros(106, dtype=np.uint64)
        self.lib.p25o_poseidon2_trace(self._p(s), self._p(tr))
        return s, tr

    def hash_no_pad(self, words):
        a = np.ascontiguousarray(words, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        self.lib.p25o_hash_no_pad(self._p(a), a.size, self._p(out))
        return out

    def merkle_commit(self, leaves_rm, cap_height, want_tree=False):
        a = np.ascontiguousarray(leaves_rm, dtype=np.uint64)
        n, w = a.shape
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        words = 0
        m = n
        while m >= (1 << cap_height):
            words += 4 * m
            if m == 1:
                break
            m >>= 1
        tree = np.zeros(words, dtype=np.uint64) if want_tree else None
        self.lib.p25o_merkle_commit(self._p(a), n, w, cap_height, self._p(cap), self._p(tree))
        return (cap, tree) if want_tree else cap

    def lde_commit(self, polys, rate_bits, cap_height, from_coeffs=False):
        a = np.ascontiguousarray(polys, dtype=np.uint64)
        npolys, n = a.shape
        log_n = n.bit_length() - 1
        coeffs = np.zeros_like(a)
        lde = np.zeros((npolys, n << rate_bits), dtype=np.uint64)
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        self.lib.p25o_lde_commit(self._p(a), log_n, npolys, int(from_coeffs), rate_bits, cap_height,
                                 self._p(coeffs), self._p(lde), self._p(cap))
        return coeffs, lde, cap


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def p25():
    return load_p25()


@pytest.fixture(scope="session")
def gpu(p25):
    p25.device_init(0)
    return p25


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
