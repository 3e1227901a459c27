"""GPU parity at BASELINE.json's full sizes (through the C ABI, against the oracle).

config 3: batch of 256 independent fib-64 verifier proofs on one GPU -- byte-compared with the oracle prover
          at the indices that straddle the 64-proof witness passes and the 16-stream context reuse
          (plonky2.5_amd/csrc/prover.hip: prove_batch_dev), oracle-verified at 16 more, deterministic.
config 5: inner Fibonacci STARK with 2^20 rows -> verifier circuit of 2^19 rows (LDE 2^22).
config 4's sharding: the REAL prover under torch.distributed (two ranks sharing GPU 0, gloo for the gather).
The reference call these replace: `data.prove(pw)`, /root/reference/src/p3/mod.rs:226-269."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu

BYTE_COMPARE = (0, 15, 16, 63, 64, 65, 255)       # witness-pass boundary (64) and stream reuse (p % 16)
VERIFY_ALSO = (1, 17, 31, 32, 47, 48, 62, 66, 127, 128, 129, 191, 192, 193, 254, 100)


def _variants(gpu, fib_inputs, count):
    """`count` distinct valid plonky3 proofs of fibonacci(64): the artifact + other PoW witnesses."""
    out = [fib_inputs]
    for v in range(1, count):
        alt, _ = gpu.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)
        assert (alt != fib_inputs).any()
        out.append(alt)
    return out


def test_config3_batch256_bytes_and_determinism(gpu, fib_circuit, fib_oracle, fib_inputs):
    B = 256
    var = _variants(gpu, fib_inputs, 8)
    batch = np.stack([var[i % 8] for i in range(B)])
    seeds = np.arange(B, dtype=np.uint64) * np.uint64(7919) + np.uint64(11)
    proofs, st = fib_circuit.prove(batch, seeds=seeds)
    assert (st == 0).all(), np.nonzero(st)[0][:8]
    # byte equality with the CPU restatement (one single-threaded oracle proof per host thread)
    idx = list(BYTE_COMPARE)
    po, sto, _per, _wall = fib_oracle.prove_many(batch[idx], seeds[idx], threads=len(idx))
    assert (sto == 0).all()
    for k, i in enumerate(idx):
        diff = np.nonzero(proofs[i] != po[k])[0]
        assert diff.size == 0, f"proof {i}: first differing words {diff[:8]}"
    dg, capg = fib_circuit.digest()
    for i in VERIFY_ALSO:
        code, msg = fib_oracle.verify(proofs[i], dg, capg)
        assert code == 0, (i, msg)
    # distinct (input, seed) -> distinct proofs; same call again -> same bytes
    assert len({proofs[i].tobytes() for i in range(B)}) == B
    proofs2, st2 = fib_circuit.prove(batch, seeds=seeds)
    assert (st2 == 0).all() and (proofs2 == proofs).all()


def test_config5_inner_trace_2_20(gpu, oracle):
    """2^20-row inner trace: 105,407 input words, outer circuit 2^19 rows x 135 wires, LDE 2^22."""
    inp, cfg = gpu.p3_prove_fibonacci(20, 100, 16, threads=os.cpu_count() or 1)
    c = gpu.Circuit.build_p3_verifier(cfg)
    assert int(c.info.degree_bits) == 19
    proofs, st = c.prove(np.stack([inp, inp]), seeds=[5, 6])
    assert st.tolist() == [0, 0]
    assert (proofs[0] != proofs[1]).any()
    oc = oracle.load_circuit(c.to_blob())
    dg, capg = c.digest()
    do, capo = oc.digest()
    assert (dg == do).all() and (capg == capo).all()
    for p in proofs:
        code, msg = oc.verify(p, dg, capg)
        assert code == 0, msg
    po, sto, _tm, msg = oc.prove(inp, seed=5)           # all host threads
    assert sto == 0, msg
    diff = np.nonzero(proofs[0] != po)[0]
    assert diff.size == 0, f"first differing proof words {diff[:8]}"
    c.close()


def test_config4_sharding_real_prover_two_ranks():
    """Two torch.distributed ranks (gloo rendezvous + gather), each running the real GPU prover on its
    shard of one batch; rank 0 checks the gathered proofs against the oracle."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), "7", "--real"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DIST_REAL_OK 7" in out.stdout


def test_config4_batch_2048_in_eight_shards_on_one_gpu(tmp_path):
    """BASELINE config 4's workload -- 2048 fib-64 verifier proofs sharded 8 x 256 -- through the HIP path on the one GPU
    there is (tests/_config4_worker.py): eight gloo ranks, each proving ITS 256 proofs of the global batch and folding them to
    its shard root (256 -> 20 -> 2 -> 1) in a GPU child process, four shard processes on the card at a time (the box admits
    six); all 2048 proofs + statuses gathered onto rank 0 (bench.py's ProofGatherer at world size 8), every status checked,
    the 16 proofs either side of the seven shard boundaries and at the two ends byte-compared with the oracle's, and the
    8-to-1 cross-rank aggregate proved over EIGHT DISTINCT shard roots: its public inputs are the commitment to all 2048
    leaves in global order and the oracle's verifier accepts it.  Path sharded: `prove(&self, ..)` borrows the circuit
    immutably, /root/reference/src/p3/mod.rs:260.  The JSON record goes to gpurun_out/ (committed under profiles/)."""
    import json
    import shutil
    import time
    out_dir, port, world = str(tmp_path), str(free_port()), 8
    ranks = []
    for r in range(world):      # the ranks started directly: torch.distributed.run would be one more process holding the GPU
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = open(os.path.join(out_dir, f"rank_{r}.log"), "w")
        ranks.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_config4_worker.py"), "ranks", "2048", out_dir,
                                        "--wave", "4", "--streams", "8"], env=env, stdout=log, stderr=subprocess.STDOUT), log))
    deadline = time.time() + 1700
    while time.time() < deadline:      # a rank that fails leaves the others in a collective: end them instead of waiting them out
        codes = [p.poll() for p, _ in ranks]
        if all(c is not None for c in codes) or any(c not in (None, 0) for c in codes):
            break
        time.sleep(0.5)
    for p, log in ranks:
        if p.poll() is None:
            p.kill()
        p.wait()
        log.close()
    codes = [p.returncode for p, _ in ranks]
    logs = [open(os.path.join(out_dir, f"rank_{r}.log")).read() for r in range(world)]
    assert codes == [0] * world, (codes, logs[0][-3000:], logs[next((i for i, c in enumerate(codes) if c), 0)][-2000:])
    assert "CONFIG4_ONE_GPU_OK 2048" in logs[0]
    rec = json.load(open(os.path.join(out_dir, "config4_one_gpu.json")))
    assert rec["total_proofs"] == 2048 and rec["ranks"] == 8 and len(rec["byte_equal_to_oracle_indices"]) == 16
    assert [s["proofs"] for s in rec["shards"]] == [256] * 8 and all(s["levels"] == [[13, 20], [10, 2], [2, 1]] for s in rec["shards"])
    assert rec["cross_rank_aggregate"]["arity"] == 8
    dst = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(dst) and os.access(dst, os.W_OK):
        shutil.copy(os.path.join(out_dir, "config4_one_gpu.json"), os.path.join(dst, "config4_one_gpu.json"))
