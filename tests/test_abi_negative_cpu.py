"""CPU: the C ABI refuses malformed input with a status code instead of crashing -- null pointers, undersized
buffers, corrupted JSON / proof bytes / blobs -- and reports P25_ERR_NO_DEVICE (never a CPU fallback) for compute
entry points on a box without a GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ARTIFACT, P


def test_json_reader_survives_corruption(p25):
    text = open(ARTIFACT, "rb").read()
    inp, cfg = p25.p3_proof_from_json(text)
    assert inp.size == 15751
    rng = np.random.default_rng(3)
    rejected = 0
    for trial in range(300):
        b = bytearray(text)
        kind = trial % 3
        if kind == 0:                                   # truncate
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:                                 # flip bytes
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:                                           # drop a slice
            i = int(rng.integers(0, len(b) - 50))
            del b[i:i + int(rng.integers(1, 50))]
        try:
            out, _cfg = p25.p3_proof_from_json(bytes(b))
            assert out.size > 0 and (out < np.uint64(P)).all()
        except p25.P25Error as e:
            assert e.status in (1, 8)                   # INVALID_ARG / PARSE
            rejected += 1
    assert rejected > 150


def test_null_and_undersized_arguments(p25):
    lib = p25.lib()
    assert lib.p25_circuit_build_gadget(0, 0, None) == 1
    assert lib.p25_circuit_import(None, 0, None) == 1
    assert lib.p25_p3_proof_from_json(None, 0, None, 0, None, None) == 1
    c = p25.Circuit.build_gadget(0, 0)
    n = C.c_size_t(0)
    assert lib.p25_circuit_export(c._h, None, 0, C.byref(n)) == 0 and n.value > 1000
    small = np.zeros(16, dtype=np.uint8)
    assert lib.p25_circuit_export(c._h, small.ctypes.data_as(C.c_void_p), small.size, C.byref(n)) == 1
    assert b"too small" in lib.p25_last_error()
    assert lib.p25_circuit_info(c._h, None) == 1
    proof = np.zeros(int(c.info.proof_words), dtype=np.uint64)
    assert lib.p25_proof_to_json(c._h, proof.ctypes.data_as(C.c_void_p), small.ctypes.data_as(C.c_void_p), 4, C.byref(n)) == 1
    assert lib.p25_proof_from_bytes(c._h, small.ctypes.data_as(C.c_void_p), small.size,
                                    proof.ctypes.data_as(C.c_void_p), proof.size) == 1
    assert lib.p25_fri_prove_words(6, 1, 2, None, 0, 3) > 0 and lib.p25_fri_prove_words(6, 1, 9, None, 0, 3) == 0
    assert lib.p25_merkle_tree_words(12, 0) == 0 and lib.p25_merkle_tree_words(8, 4) == 0


def test_no_device_means_error_not_fallback(p25):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = p25.lib()
    st = np.zeros(12, dtype=np.uint64)
    assert lib.p25_poseidon_permute(st.ctypes.data_as(C.c_void_p), 1) == 2          # P25_ERR_NO_DEVICE
    c = p25.Circuit.build_gadget(0, 0)
    with pytest.raises(p25.P25Error) as e:
        c.prove(np.zeros(3, dtype=np.uint64))
    assert e.value.status == 2 and "no CPU fallback" in str(e.value)
    with pytest.raises(p25.P25Error):
        c.build_recursive_verifier(1)          # needs the device for the inner digest unless it is passed in
    with pytest.raises(p25.P25Error):
        p25.transcript([([1, 2, 3], 1)])


def test_blob_beyond_kernel_capacities_is_rejected(p25):
    """A blob the header checks alone would accept but the device kernels cannot hold: rate_bits = 2 with
    80 routed wires means 20 partial-product chunks per challenge (k_zpp_chunks holds MAX_CHUNKS = 16 in registers) and
    gate constraints of degree above the quotient's -- refused at import, before any kernel sees it."""
    import struct
    blob = bytearray(p25.Circuit.build_gadget(0, 0).to_blob())
    hdr = list(struct.unpack_from("<32Q", blob, 8))
    assert hdr[5] == 8 and hdr[6] == 3 and hdr[2] == 80 and hdr[13] == 9   # qdf, rate_bits, routed wires, NP
    hdr[5], hdr[6], hdr[13] = 4, 2, 19                                      # consistent among themselves
    struct.pack_into("<32Q", blob, 8, *hdr)
    with pytest.raises(p25.P25Error) as e:
        p25.Circuit.from_blob(bytes(blob))
    assert e.value.status == 1 and "MAX_CHUNKS" in str(e.value)
    # rate_bits 2 with few enough routed wires for the chunk arrays still fails: the gate set's degree needs rate_bits 3
    hdr[2], hdr[13] = 64, 15
    struct.pack_into("<32Q", blob, 8, *hdr)
    with pytest.raises(p25.P25Error) as e:
        p25.Circuit.from_blob(bytes(blob))
    assert e.value.status == 1


def test_runtime_info_and_comm_entry_points_without_a_gpu(p25):
    """Round 6's entry points on a box without a GPU: p25_runtime_info answers (it touches no device), p25_device_init_ex
    validates its arguments before anything else, and the communicator calls report P25_ERR_NO_DEVICE -- never a crash, never a
    CPU stand-in for the collective."""
    lib = p25.lib()
    ri = p25.runtime_info().as_dict()
    assert ri["proving_streams"] == 16 and ri["main_streams"] == 2 and ri["hw_queues_setting_late"] == 0
    assert lib.p25_runtime_info(None) == 1
    assert lib.p25_device_init_ex(0, -1) == 1 and lib.p25_device_init_ex(0, 1000) == 1       # before any HIP call
    st = lib.p25_device_init_ex(0, 0)
    assert st in (0, 2)                                  # 2 = no device here; hw_queues 0 leaves the environment alone
    buf = (C.c_uint8 * 128)()
    assert lib.p25_comm_unique_id(buf) in (2, 10)        # NO_DEVICE (or RCCL absent)
    h = C.c_void_p()
    assert lib.p25_comm_init(buf, 0, 1, C.byref(h)) in (2, 10) and not h.value
    assert lib.p25_comm_rank(None) == -1 and lib.p25_comm_world(None) == 0 and not lib.p25_comm_stream(None)
    assert lib.p25_comm_destroy(None) == 0
    cnt = (C.c_size_t * 1)(0)
    assert lib.p25_gather_proofs(None, None, -1, None, 1, None, cnt, 0, None, None) in (1, 2, 10)
    with pytest.raises(p25.P25Error):
        p25.comm_unique_id()


def test_p3_prove_air_ex_argument_checks(p25):
    """log_blowup out of range, a degree the blow-up cannot hold, a shape whose Merkle-path length disagrees with its FriConfig."""
    import air_cases
    air, par = air_cases.quartic_map(p25, 5)
    trace = air_cases.quartic_map_trace(par, 3)
    for lb in (0, 1, 5):
        with pytest.raises(p25.P25Error) as e:
            p25.p3_prove_air(air, trace, num_queries=3, pow_bits=3, log_blowup=lb)
        assert e.value.status == 1
    inp, cfg = p25.p3_prove_air(air, trace, num_queries=3, pow_bits=3, log_blowup=2)
    bad = p25.P3Config(2, 3, 3, 2, 3, 2, 4, 2, 3)       # opening_matrix_log_max_height 4 != log_trace_height 3 + log_blowup 2
    with pytest.raises(p25.P25Error):
        p25.Circuit.build_p3_verifier_air(bad, air)
    assert int(p25.Circuit.build_p3_verifier_air(cfg, air).info.num_inputs) == inp.size
