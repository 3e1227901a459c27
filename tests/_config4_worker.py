"""BASELINE config 4's WORKLOAD on a one-GPU box: a batch of 2048 fib-64 verifier proofs in 8 shards of 256, one rank
per shard, the finished proofs gathered onto rank 0, every shard folded to one root and the 8 DISTINCT shard roots folded
once more (tests/test_gpu_baseline_configs.py::test_config4_batch_2048_in_eight_shards_on_one_gpu).

A GPU box admits at most 6 processes on its card, so the 8 ranks cannot all hold the GPU.  The job is therefore:

  RANK=r WORLD_SIZE=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/_config4_worker.py ranks TOTAL OUTDIR [--wave 4] [--streams 8]
      (once per rank; the test starts the eight itself -- `python -m torch.distributed.run` opens the GPU in the launcher)
      eight gloo ranks that never touch the GPU themselves (measured on the box, tools/probe/: importing torch, gloo init,
      all_reduce, all_gather_object and the gatherer leave /dev/kfd closed; `dist.barrier()` opens it -- torch picks a device for
      it -- so the ranks synchronise with an all_reduce instead).  Rank r runs ITS shard's GPU work -- p25_prove_batch over proofs
      [256 r, 256 r + 256) of the global batch, then the shard's aggregation tree 256 -> 20 -> 2 -> 1 -- in a CHILD process
      (`shard` mode below), `--wave` ranks at a time (barrier between the waves: 4 children + the test's own process on the
      card at once); then the collective path of bench.py at world size 8: `ProofGatherer` over all 2048 proofs + statuses,
      one more over the 8 shard roots.  Rank 0 then takes the GPU itself for the 8-to-1 cross-rank aggregate and checks
      everything with the oracle (the checker): all 2048 statuses, 16 gathered proofs straddling the shard boundaries byte
      for byte, the root's public inputs == expected_commitment over all 2048 leaves, the oracle's verifier on the root.

  python tests/_config4_worker.py shard RANK WORLD TOTAL OUTDIR [--streams 8]
      one shard's GPU work in a process of its own; writes OUTDIR/shard_RANK.npz.

Only the transport differs from an 8-GPU run: gloo on host copies instead of RCCL (RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_VARIANTS = 11      # distinct plonky3 proofs cycled through the global batch; coprime to 256: every shard starts elsewhere
SEED0 = 7000


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def global_batch(p25, start, stop):
    """Rows [start, stop) of the global batch: inputs and filler seeds are functions of the GLOBAL proof index."""
    with open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")) as f:
        base, cfg = p25.p3_proof_from_json(f.read())
    need = sorted({g % N_VARIANTS for g in range(start, stop)})
    var = {0: base}
    for v in need:
        if v:
            var[v] = p25.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24, threads=2)[0]
    rows = np.stack([var[g % N_VARIANTS] for g in range(start, stop)])
    seeds = np.arange(start, stop, dtype=np.uint64) + np.uint64(SEED0)
    return rows, seeds, cfg


def main_shard(rank, world, total, outdir):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    import __graft_entry__ as ge
    p25 = ge.load_package()
    from plonky25_amd import aggregate as ag, dist as pd
    streams = arg("--streams", 8)
    late = p25.device_init(0)
    start, stop = pd.shard_range(total, rank, world)
    rows, seeds, cfg = global_batch(p25, start, stop)
    circuit = p25.Circuit.build_p3_verifier(cfg)
    circuit.set_streams(streams)       # several shard processes share one GPU's memory here: fewer working sets each
    circuit.digest()
    circuit.prove(rows[:1], seeds=seeds[:1])          # contexts and tables, like the warm-up steps of bench.py
    t = time.perf_counter()
    proofs, st = circuit.prove(rows, seeds=seeds)
    prove_s = time.perf_counter() - t
    rec = {"rank": rank, "proofs": int(stop - start), "prove_s": round(prove_s, 3), "hw_queues_setting_late": bool(late),
           "proofs_per_s_time_shared": round((stop - start) / prove_s, 2), "statuses_ok": bool((st == 0).all())}
    root = np.zeros(0, dtype=np.uint64)
    if rec["statuses_ok"]:
        arity = arg("--arity", 13)     # aggregate.widest_arity(circuit) for this circuit (tests/test_gpu_recursion.py pins it)
        t = time.perf_counter()
        f = ag.fold(circuit, [proofs[i] for i in range(proofs.shape[0])], arity=arity, warm=False, streams=streams)
        rec.update({"arity": arity, "fold_s": round(time.perf_counter() - t, 3), "tree_prove_s": round(f["tree_s"], 3),
                    "levels": [(l["arity"], l["proofs"]) for l in f["levels"]],
                    "root_public_inputs": [int(v) for v in f["top"].public_inputs(f["root"])]})
        root = f["root"]
    np.savez(os.path.join(outdir, f"shard_{rank}.npz"), proofs=proofs, status=st.astype(np.int32), root=root,
             record=np.frombuffer(json.dumps(rec).encode(), dtype=np.uint8))
    print("SHARD_DONE", json.dumps(rec), flush=True)


def main_ranks(total, outdir):
    import datetime
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=1700))
    rank, world = dist.get_rank(), dist.get_world_size()
    p25 = ge.load_package()
    from plonky25_amd import aggregate as ag, dist as pd
    cpu = torch.device("cpu")
    wave, streams = arg("--wave", 4), arg("--streams", 8)

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    t_job = time.perf_counter()
    for w in range(-(-world // wave)):
        ok = True
        if rank // wave == w:      # this rank's GPU work, in a child: at most `wave` shard processes on the card at once
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "shard", str(rank), str(world), str(total), outdir,
                                "--streams", str(streams), "--arity", str(arg("--arity", 13))], capture_output=True, text=True, timeout=1500)
            ok = r.returncode == 0 and "SHARD_DONE" in r.stdout
            if not ok:
                sys.stderr.write(f"rank {rank}: shard process failed\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}\n")
        if not agree(ok):
            sys.exit(3)
    gpu_s = time.perf_counter() - t_job
    z = np.load(os.path.join(outdir, f"shard_{rank}.npz"))
    proofs, st, root = z["proofs"], z["status"], z["root"]
    rec = json.loads(bytes(z["record"]).decode())
    pw = int(proofs.shape[1])
    if not agree(root.size > 0):
        sys.exit(4)
    # the final aggregation step at world size 8: all 2048 proofs + statuses onto rank 0 (bench.py's gatherer), then the roots
    t = time.perf_counter()
    g = pd.ProofGatherer(total, pw, cpu)
    blocks, sts = g.gather(torch.from_numpy(proofs.view(np.int64)), torch.from_numpy(st))
    gather_s = time.perf_counter() - t
    rg = pd.ProofGatherer(world, int(root.size), cpu)
    r_blocks, _ = rg.gather(torch.from_numpy(root.view(np.int64).copy())[None, :], torch.zeros(1, dtype=torch.int32))
    recs = [None] * world
    dist.all_gather_object(recs, rec)
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_binding import Oracle      # the checker
        ora = Oracle()
        all_p = np.concatenate([b.numpy().view(np.uint64) for b in blocks])
        all_s = np.concatenate([s.numpy() for s in sts])
        assert all_p.shape == (total, pw) and all_s.shape == (total,)
        assert (all_s == 0).all(), np.nonzero(all_s)[0][:16].tolist()
        sizes = pd.shard_sizes(total, world)
        assert [int(b.shape[0]) for b in blocks] == sizes
        # the gathered batch is the global batch in order: every proof distinct (its filler seed is its global index)
        caps = all_p[:, :ag.CAP_WORDS]
        assert len({c.tobytes() for c in caps}) == total, "two proofs of the batch share a wires cap"
        # rank 0 takes the GPU now (the shard processes are gone): the leaf circuit, the shard trees' circuits, the 8-to-1 top
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
        p25.device_init(0)
        rows, seeds, cfg = global_batch(p25, 0, 1)
        circuit = p25.Circuit.build_p3_verifier(cfg)
        circuit.set_streams(streams)
        dg, cs_cap = circuit.digest()
        oc = ora.load_circuit(circuit.to_blob())
        # 16 gathered proofs straddling the shard boundaries (last of shard r, first of shard r + 1, the two ends), byte for byte
        bounds = np.cumsum(sizes)[:-1]
        idx = sorted({0, total - 1} | {int(b) - 1 for b in bounds} | {int(b) for b in bounds})
        want_rows = np.stack([global_batch(p25, i, i + 1)[0][0] for i in idx])
        want_seeds = np.array([i + SEED0 for i in idx], dtype=np.uint64)
        ora.set_tuned(True)        # AVX-512 legs where the host has them (validated against the scalar oracle in the CPU suite)
        t = time.perf_counter()
        po, sto, _per, _wall = oc.prove_many(want_rows, want_seeds, threads=min(8, len(idx), os.cpu_count() or 1))
        oracle_s = time.perf_counter() - t
        ora.set_tuned(False)
        assert (sto == 0).all()
        for k, i in enumerate(idx):
            assert (all_p[i] == po[k]).all(), f"gathered proof {i} differs from the oracle's"
        for i in range(0, total, max(1, total // 16)):        # the oracle's verifier, spread over the whole batch
            code, msg = oc.verify(all_p[i], dg, cs_cap)
            assert code == 0, (i, msg)
        # eight DISTINCT shard roots -> one proof
        roots = [b[0].numpy().view(np.uint64) for b in r_blocks]
        assert len({r.tobytes() for r in roots}) == world, "two shards produced the same root"
        arity = recs[0]["arity"]
        assert all(r["arity"] == arity and r["levels"] == recs[0]["levels"] for r in recs)
        circ, level_circs = circuit, []
        for k, _n in recs[0]["levels"]:
            circ = circ.build_aggregator(k)
            circ.digest()
            circ.set_streams(2)
            level_circs.append(circ)
        for q in range(world):      # every shard root carries the commitment to ITS 256 leaves
            lo = sum(sizes[:q])
            assert recs[q]["root_public_inputs"] == ag.expected_commitment(list(caps[lo:lo + sizes[q]]), arity, ora.hash_no_pad), q
        t = time.perf_counter()
        fin = ag.fold_roots(circ, roots, warm=False)
        cross_s = time.perf_counter() - t
        got = [int(v) for v in fin["top"].public_inputs(fin["root"])]
        want = ag.expected_commitment(list(caps), arity, ora.hash_no_pad, n_shards=world)
        assert got == want, (got, want)
        tdg, tcap = fin["top"].digest()
        code, msg = ora.load_circuit(fin["top"].to_blob()).verify(fin["root"], tdg, tcap)
        assert code == 0, msg
        out = {"workload": f"BASELINE config 4 on ONE GPU: {total} fib-64 verifier proofs in {world} shards (one gloo rank each, the "
                           f"rank's GPU work in a child process, {wave} shard processes on the card at a time), gathered onto rank 0, "
                           f"every shard folded {recs[0]['levels']} and the {world} distinct shard roots folded to one proof",
               "total_proofs": total, "ranks": world, "all_statuses_ok": True, "distinct_wires_caps": total,
               "byte_equal_to_oracle_indices": idx, "oracle_prove_s": round(oracle_s, 1),
               "root_public_inputs": got, "root_public_inputs_commit_to_all_leaves": True, "oracle_verifier_accepts_root": True,
               "cross_rank_aggregate": fin["levels"][0], "cross_rank_s_incl_build": round(cross_s, 2),
               "gather_of_all_proofs_s_gloo_host": round(gather_s, 3), "gpu_phase_s_all_waves": round(gpu_s, 1),
               "proofs_per_s_whole_job_time_shared_gpu": round(total / gpu_s, 1),
               "shards": recs}
        with open(os.path.join(outdir, "config4_one_gpu.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("CONFIG4_ONE_GPU_OK", total, flush=True)
    agree(True)       # NOT dist.barrier(): that opens the GPU in every rank (nine processes on the card: the box kills the run)
    dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "shard":
        main_shard(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        main_ranks(int(sys.argv[2]), sys.argv[3])
