#!/usr/bin/env python3
"""bench.py -- recursive proofs/s for the fib-64 plonky3-in-plonky2 verifier circuit on MI355X.

One "step" = one pass of the hot path (`data.prove(pw)`, /root/reference/src/p3/mod.rs:260) over a
batch of `--batch` independent proofs per GPU (BASELINE.json configs[2]/[3]: 256 proofs per GPU),
inputs resident in HBM before the timed region.  Proofs are independent, so N GPUs = N replicas of
the circuit tables, each proving its own shard (weak scaling); the only collective is the RCCL
gather of the finished proofs onto rank 0 at the end of each step.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- the dominant kernel (Poseidon leaf sponge over the 2^19 x 135 wires LDE): algorithmic
                  bytes per launch / its measured duration (HIP events on the proving stream) vs 8 TB/s
  cpu_baseline -- the oracle (CPU restatement, kind "port") proving the same input on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

# libp25 sets this itself when it is loaded; torch may initialise HIP first, so set it here too
# (hardware queues the runtime spreads the prover's streams over -- see capi.hip).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="proofs per GPU per step")
    ap.add_argument("--log-n", type=int, default=6, help="log2 rows of the inner Fibonacci STARK (6 = the artifact)")
    ap.add_argument("--distinct", type=int, default=8, help="number of distinct plonky3 proofs cycled through the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="oracle threads (0 = all host cores)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    p25 = ge.load_package()
    p25.device_init(local_rank)
    import p3json
    if args.log_n == 6:
        inputs, _shape = p3json.load(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json"))
        p3cfg = p25.P3Config.fib64()
    else:  # BASELINE config 5 and friends: inner STARK with 2^log_n rows from the native plonky3 prover
        inputs, p3cfg = p25.p3_prove_fibonacci(args.log_n, 100, 16, threads=os.cpu_count() or 1)
    # distinct batch items: further valid plonky3 proofs of the same statement (other PoW witnesses ->
    # other query indices); the reference artifact is item 0 for log_n = 6
    variants = [inputs]
    for v in range(1, max(1, min(args.distinct, args.batch))):
        alt, _ = p25.p3_prove_fibonacci(args.log_n, 100, 16, pow_start=(v << 24) + rank * (1 << 20),
                                        threads=os.cpu_count() or 1)
        variants.append(alt)

    # circuit: built once per shape by the host code, tables made resident on the GPU
    t0 = time.time()
    circuit = p25.Circuit.build_p3_verifier(p3cfg)
    info = circuit.info
    digest, cs_cap = circuit.digest()  # forces the device-side tables (constants/sigmas commitment)
    build_s = time.time() - t0
    B = args.batch
    ni, pw = int(info.num_inputs), int(info.proof_words)
    dev = torch.device("cuda", local_rank)
    host_in = np.stack([variants[i % len(variants)] for i in range(B)]).view(np.int64)
    d_inputs = torch.from_numpy(host_in).to(dev)                                           # [B][ni]
    d_seeds = (torch.arange(B, dtype=torch.int64) + rank * B).to(dev)                      # distinct filler seeds
    d_proofs = torch.zeros((B, pw), dtype=torch.int64, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # inputs are resident in HBM before anything is timed
    from plonky25_amd import dist as pdist

    def step():
        circuit.prove_dev(d_inputs.data_ptr(), B, d_seeds.data_ptr(), d_proofs.data_ptr(), pw, d_status.data_ptr())
        circuit.sync()
        if distributed:  # the final aggregation step: finished proofs gathered onto rank 0 over RCCL/xGMI
            pdist.gather_proofs(d_proofs, d_status, world * B)

    for _ in range(args.warmup):
        step()
    circuit.kernel_stats(enable=True, reset=True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms, k_launches = circuit.kernel_stats(enable=False, reset=False)

    statuses = d_status.cpu().numpy()
    ok = bool((statuses == 0).all())
    if distributed:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    if rank == 0:
        # per-phase device times of one proof (outside the timed region)
        _p, _s, tm = circuit.prove(inputs, seeds=[0], timings=True)
        # correctness outside the timed region: the oracle verifier accepts a proof of the last batch
        from conftest import Oracle
        ora = Oracle()
        oc = ora.load_circuit(circuit.to_blob())
        proof0 = d_proofs[0].cpu().numpy().view(np.uint64)
        vcode, vmsg = oc.verify(proof0, digest, cs_cap)
        n_big = 1 << (int(info.degree_bits) + 3)
        algo_bytes = n_big * int(info.num_wires) * 8 + n_big * 32   # read the LDE once, write one digest per leaf
        avg_ms = k_ms / max(1, k_launches)
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_hash_leaves.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        total_proofs = world * B * args.steps
        # Integer-VALU view of the same run (the bound that actually binds): wave-level VALU instructions
        # per proof from the latest committed PMC pass (SQ_INSTS_VALU, profiles/*_pmc_SQ_INSTS_VALU.json,
        # written by tools/collect_profiles.sh) x proofs/s per GPU, against one VALU instruction per 4 cycles
        # per SIMD (1024 SIMDs, 2.4 GHz).
        valu = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_SQ_INSTS_VALU.json")))
        vp = cands[-1] if cands else ""
        if vp and args.log_n == 6:
            try:
                per_kernel = json.load(open(vp))
                instr_per_proof = sum(v.get("SQ_INSTS_VALU", 0.0) for v in per_kernel.values())
                peak = 1024 * 2.4e9 / 4
                ach = instr_per_proof * (total_proofs / elapsed) / world
                valu = {"wave_instr_per_proof": instr_per_proof, "achieved_wave_instr_per_s": ach,
                        "peak_wave_instr_per_s": peak, "frac": ach / peak,
                        "leaf_hash_share": sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in per_kernel.items()
                                               if "k_hash_leaves" in k) / instr_per_proof,
                        "note": "peak = 1 VALU instruction / SIMD / 4 cycles at 2.4 GHz (shader clock measured in-kernel "
                                "under this load: 2.31-2.40 GHz); v_mad_u64_u32, 60% of the mix, issues every ~4.7 "
                                "cycles, so the practical ceiling is ~0.88"}
            except Exception:
                valu = None
        # HBM view per phase and overall: algorithmic bytes of each phase -- inputs read
        # once, outputs written once -- over that phase's device time for one proof alone on the GPU, and
        # all phases x proofs/s for the batch run.
        n_small, nw = 1 << int(info.degree_bits), int(info.num_wires)
        nr, ncs = int(info.num_routed_wires), int(info.num_constants_sigmas)
        npp = -(-nr // 8) - 1          # chunks of 8 routed wires (max quotient degree factor) minus the Z column
        nz = 2 * (1 + npp)
        W = 8
        ntt = lambda cols: cols * (2 * n_small + n_big) * W           # read values, write coeffs, write LDE
        mrk = lambda cols: n_big * cols * W + 2 * n_big * 32            # read LDE, write digests + inner levels
        phase_bytes = {
            "witness_ms": n_small * nw * W,
            "wires_commit_ms": ntt(nw) + mrk(nw),
            "partial_products_ms": n_small * 2 * nr * W + n_small * nz * W,
            "zs_commit_ms": ntt(nz) + mrk(nz),
            "quotient_ms": n_big * (nw + ncs + nz + 2) * W + n_big * 2 * W,
            "quotient_commit_ms": 2 * n_big * 2 * W + ntt(16) + mrk(16),
            "openings_ms": (nw + ncs + nz + 16) * n_small * W,
            "fri_ms": (nw + ncs + nz + 16) * n_small * W + 30e6,
        }
        tmd = tm.as_dict()
        hbm_phases = {k[:-3]: {"alg_MB": round(b / 1e6, 1), "GBps_single_proof": round(b / (tmd[k] * 1e-3) / 1e9, 1)}
                      for k, b in phase_bytes.items() if tmd.get(k, 0) > 0}
        bytes_per_proof = float(sum(phase_bytes.values()))
        hbm_overall = {"alg_GB_per_proof": round(bytes_per_proof / 1e9, 3),
                       "achieved_GBps": round(bytes_per_proof * (total_proofs / elapsed) / world / 1e9, 1),
                       "frac_of_peak": round(bytes_per_proof * (total_proofs / elapsed) / world / 1e9 / HBM_PEAK_GBS, 4)}
        out = {
            "metric": "recursive proofs/sec (fib-64 p3-in-p2 circuit)",
            "value": total_proofs / elapsed,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, p = 2^64 - 2^32 + 1)",
            "data": f"synthetic: {len(variants)} distinct plonky3 proofs of fibonacci(2^{args.log_n}) "
                    "(item 0 = the reference's artifacts/proof_fibonacci.json for log_n 6; the others from the native "
                    "p3 prover with other PoW witnesses) cycled through the batch, distinct filler seeds",
            "config": {"workload": f"batch of {B} independent fib-64 plonky3-verifier proofs per GPU "
                                   f"(n = 2^{int(info.degree_bits)} rows x 135 wires, LDE 2^{int(info.degree_bits) + 3}), "
                                   f"{world} GPU(s), replicas + RCCL gather",
                       "proofs_per_gpu_per_step": B, "all_statuses_ok": ok,
                       "oracle_verifier_accepts": vcode == 0, "circuit_build_s": round(build_s, 2),
                       "phase_ms_single_proof": {k: round(v, 3) for k, v in tm.as_dict().items()}},
            "roofline": {"bound": "hbm", "kernel": "k_hash_leaves_wide (Poseidon sponge, 2^19 leaves x 135 words)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": int(k_launches), "algorithmic_bytes": algo_bytes,
                         "note": "integer-VALU bound (17 Poseidon permutations per leaf), not HBM bound; "
                                 "avg_launch_ms is measured with 16 proofs in flight sharing the GPU "
                                 "(3.6 ms when the kernel runs alone)",
                         "valu": valu, "hbm_phases": hbm_phases, "hbm_overall": hbm_overall},
        }
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0, N = 1 only
            threads = args.cpu_threads or (os.cpu_count() or 1)
            ora.set_threads(threads)
            oc.digest()  # constants/sigmas commitment is per-circuit, excluded like the reference's build()
            t0 = time.perf_counter()
            _pr, st, otm, msg = oc.prove(inputs, seed=0)
            cpu_s = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "proofs/s", "cores": threads, "kind": "port",
                                   "sample": f"1 full fib-64 proof (witness + prove) by the oracle C++ restatement, "
                                             f"{threads} threads, {cpu_s:.1f} s; status {st}",
                                   "phase_s": {k: round(v, 2) for k, v in otm.items()}}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
