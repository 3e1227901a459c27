"""Flatten a plonky3 proof JSON (format of /root/reference/src/p3/serde/proof.rs:16-355, e.g. the
reference's artifacts/proof_fibonacci.json, kept as a data fixture under tests/golden/) into the
prover's per-proof input vector, in `Proof::<Target>::add_virtual_to` order (proof.rs:357-373)."""
import json

import numpy as np


def _v(x):
    return int(x["value"])


def _ext(e):
    return [_v(x) for x in e["value"]]


def flatten_p3_proof(obj):
    out = []
    out += [_v(x) for x in obj["commitments"]["trace"]["value"]]
    out += [_v(x) for x in obj["commitments"]["quotient_chunks"]["value"]]
    ov = obj["opened_values"]
    for e in ov["trace_local"]:
        out += _ext(e)
    for e in ov["trace_next"]:
        out += _ext(e)
    assert len(ov["quotient_chunks"]) in (1, 2, 4, 8)  # proof.rs:41-48 hard-codes one chunk; 2 / 4 / 8 = the degree 3 / 4-5 / 6-9 extensions
    for chunk in ov["quotient_chunks"]:
        for e in chunk:
            out += _ext(e)
    fp = obj["opening_proof"]["fri_proof"]
    for c in fp["commit_phase_commits"]:
        out += [_v(x) for x in c["value"]]
    for qp in fp["query_proofs"]:
        for step in qp["commit_phase_openings"]:
            out += _ext(step["sibling_value"])
            for sib in step["opening_proof"]:
                out += [_v(x) for x in sib]
    out += _ext(fp["final_poly"])
    out.append(_v(fp["pow_witness"]))
    for qo in obj["opening_proof"]["query_openings"]:
        for batch in qo:
            for row in batch["opened_values"]:
                out += [_v(x) for x in row]
            for sib in batch["opening_proof"]:
                out += [_v(x) for x in sib]
    return np.array(out, dtype=np.uint64)


def p3_shape(obj):
    """P3Config derived from the proof's shape (src/p3/mod.rs:74-87)."""
    n_chunks = len(obj["opened_values"]["quotient_chunks"])
    return {
        "log_quotient_degree": max(0, (n_chunks - 1).bit_length()),
        "log_trace_height": len(obj["opening_proof"]["fri_proof"]["commit_phase_commits"]),
        "trace_width": len(obj["opened_values"]["trace_local"]),
        "opening_matrix_log_max_height": len(obj["opening_proof"]["query_openings"][0][0]["opening_proof"]),
        "quotient_opened_len": len(obj["opening_proof"]["query_openings"][0][1]["opened_values"][0]),
        "degree_bits": obj["degree_bits"],
        "num_queries": len(obj["opening_proof"]["fri_proof"]["query_proofs"]),
        # FriConfig.log_blowup is not in the proof; the input Merkle paths are log_trace_height + log_blowup long (verifier.rs:264)
        "log_blowup": len(obj["opening_proof"]["query_openings"][0][0]["opening_proof"])
                      - len(obj["opening_proof"]["fri_proof"]["commit_phase_commits"]),
    }


def load(path):
    obj = json.load(open(path))
    return flatten_p3_proof(obj), p3_shape(obj)
