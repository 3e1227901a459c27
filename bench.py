#!/usr/bin/env python3
"""bench.py -- recursive proofs/s for the fib-64 plonky3-in-plonky2 verifier circuit on MI355X.

One "step" = one pass of the hot path (`data.prove(pw)`, /root/reference/src/p3/mod.rs:260) over a
batch of `--batch` independent proofs per GPU (BASELINE.json configs[2]/[3]: 256 proofs per GPU),
inputs resident in HBM before the timed region.  Proofs are independent, so N GPUs = N replicas of
the circuit tables, each proving its own shard (weak scaling); the only collective is the RCCL
gather of the finished proofs onto rank 0 at the end of each step.

`python bench.py --gpus N` works bare: with N > 1 and no RANK in the environment it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process BEFORE torch or HIP
is touched, and exits with the child's code.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- the dominant kernel (Poseidon leaf sponge over the 2^19 x 135 wires LDE): algorithmic
                  bytes per launch / its duration measured with HIP events on the proving stream, for
                  launches that have the GPU to themselves (the figure rocprofv3 reports per kernel) and,
                  separately, for the launches of the timed region (16 proofs in flight, time-sliced)
  cpu_baseline -- the oracle (CPU restatement, kind "port") on the host cores: one proof on ONE pinned thread (the
                  reference build has no `parallel` feature, Cargo.toml:15-18) and the whole host (independent proofs
                  in flight on all physical cores, 16 threads each).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s
COLLECTIVE_TIMEOUT_S = 1800   # torch.distributed process-group timeout (RCCL watchdog included)
N_SIMD, NOMINAL_CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; the clock is measured in the run (p25_shader_clock_hz)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="proofs per GPU per step")
    ap.add_argument("--log-n", type=int, default=6, help="log2 rows of the inner Fibonacci STARK (6 = the artifact)")
    ap.add_argument("--distinct", type=int, default=8, help="number of distinct plonky3 proofs cycled through the batch")
    ap.add_argument("--verify", type=int, default=8, help="proofs of the last step checked by the oracle verifier")
    ap.add_argument("--cpu-baseline", choices=("full", "one-thread", "none"), default="full")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline none")
    ap.add_argument("--cpu-cores", type=int, default=0, help="physical cores of the whole-host leg (0 = all)")
    ap.add_argument("--total", type=int, default=0,
                    help="strong-scaling mode (BASELINE config 4 literally: --total 2048): this many proofs per step in "
                         "all, split across the ranks; overrides --batch")
    ap.add_argument("--aggregate", type=int, default=-1,
                    help="leaves of the aggregation-tree measurement after the timed region (recursive k-to-1 verifier "
                         "circuits down to ONE root proof, k = --aggregate-arity), PER RANK: every rank folds that many proofs of its own shard to one "
                         "root, the N roots are gathered and rank 0 proves one N-to-1 aggregate on top; -1 = 64 (0 = off)")
    ap.add_argument("--aggregate-arity", type=int, default=0, choices=[0] + list(range(2, 17)), metavar="0 | 2..16",
                    help="children per aggregation circuit, at most (plonky25_amd.aggregate.level_plan); 0 = the arity at which "
                         "an aggregation circuit costs least per child (aggregate.widest_arity).  For fib-64 verifier proofs "
                         "that is 13, the most the circuit's 2^16 rows hold (62,753 rows; 8 use 38,687, 14 need 2^17): the same "
                         "machine time per aggregate proof, 23 instead of 37 of them per 256 leaves")
    ap.add_argument("--extra-configs", choices=("auto", "none"), default="auto",
                    help="auto: also measure BASELINE configs 2 (single proof) and 5 (2^20-row inner STARK) and report "
                         "them in the `configs` block (N = 1 only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the distributed branch even with --gpus 1: a world-size-1 RCCL process group on cuda:0, so "
                         "init, the device-tensor gather and the float64 collectives execute on a one-GPU box")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL over xGMI, one GPU per rank (the real thing); gloo = test mode: every rank on GPU 0, "
                         "collectives on host copies (exercises the multi-rank path on a one-GPU box)")
    ap.add_argument("--dist-native", action="store_true",
                    help="the gather of the finished proofs through the library's own communicator (p25_comm_init / "
                         "p25_gather_proofs: librccl called from libp25 on a library-owned side stream, include/p25.h) instead of "
                         "torch.distributed -- what a Rust host runs; torch.distributed stays the launcher's out-of-band channel "
                         "(unique id, barriers, the timing reductions).  Needs --dist-backend nccl")
    return ap.parse_args()


def fail(msg, code=2):
    """One JSON line saying why there is no measurement (the driver reads stdout), then a non-zero exit."""
    print(json.dumps({"metric": "recursive proofs/sec (fib-64 p3-in-p2 circuit)", "value": None, "unit": "proofs/s",
                      "error": msg}), flush=True)
    sys.exit(code)


def visible_gpus(in_child=False):
    """GPUs torch can use.  `in_child`: asked by the launcher parent, which must stay free of torch and HIP -- the count
    is taken by a short-lived child process (sysfs/KFD topology would also list GPUs this container cannot open)."""
    if in_child:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True)
        try:
            return int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            return 0
    import torch  # counting devices does not initialise HIP on this image
    return torch.cuda.device_count()


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: become the launcher.  This process has not imported torch or touched HIP
    (the GPU count came from a child process), and the ranks are a CHILD process too, never an exec of this one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def physical_cores():
    """One logical CPU per physical core of this process's affinity set (SMT siblings dropped)."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                key = f.read().strip()
        except OSError:
            key = str(c)
        if key not in seen:
            seen.add(key)
            out.append(c)
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measured_clock(p25, torch, dev):
    """(Hz, how it was obtained): the shader clock under a full-chip Poseidon load, from the in-kernel cycle counter against
    the constant-rate wall clock (p25_shader_clock_hz).  A reading is accepted only between half of and 5 % above the
    device's rated maximum (a box has been seen to return 3.17 GHz for a 2.4 GHz part, another 2.09 GHz right after an idle
    stretch); the median of five readings after two warm-up readings is used, the rated clock if fewer than three are
    plausible, and the note says which."""
    try:
        rated = float(torch.cuda.get_device_properties(dev).clock_rate) * 1e3     # kHz -> Hz
    except Exception:
        rated = 0.0
    if not (1.0e9 < rated < 3.5e9):
        rated = NOMINAL_CLOCK_HZ
    good, err = [], ""
    for attempt in range(7):     # the first two readings warm the clocks up (a probe right after an idle stretch reads low)
        try:
            hz = p25.shader_clock_hz()
        except Exception as e:
            hz, err = 0.0, str(e)[:120]
        if attempt < 2:
            continue
        if 0.5 * rated < hz < 1.05 * rated:
            good.append(hz)
        else:
            err = err or f"implausible reading {hz:.4g} Hz against a rated {rated:.4g} Hz"
    if len(good) >= 3:
        good.sort()
        return good[len(good) // 2], ("measured in this run: median of %d readings %.3f-%.3f GHz (in-kernel cycle counter vs the "
                                      "constant-rate wall clock, full-chip Poseidon load)" % (len(good), good[0] / 1e9, good[-1] / 1e9))
    return rated, f"rated maximum (the in-run measurement gave fewer than three plausible readings: {err})"


def bench_config5(p25, np, torch, dev, host_threads, verify):
    """BASELINE config 5: inner STARK = Fibonacci trace of 2^20 rows (outer circuit 2^19 rows, LDE 2^22), 1 GPU."""
    t = time.perf_counter()
    inp, cfg = p25.p3_prove_fibonacci(20, 100, 16, threads=host_threads)
    alt, _ = p25.p3_prove_fibonacci(20, 100, 16, pow_start=1 << 24, threads=host_threads)
    p3_s = time.perf_counter() - t
    t = time.perf_counter()
    circ = p25.Circuit.build_p3_verifier(cfg)
    dg, cap = circ.digest()
    build_s = time.perf_counter() - t
    info = circ.info
    B, pw = 16, int(info.proof_words)
    host_in = np.stack([inp if i % 2 == 0 else alt for i in range(B)])
    d_in = torch.from_numpy(host_in.view(np.int64)).to(dev)
    d_seeds = torch.arange(B, dtype=torch.int64, device=dev)
    d_proofs = torch.zeros((B, pw), dtype=torch.int64, device=dev)
    steps = 3
    d_status = torch.zeros((1 + steps, B), dtype=torch.int32, device=dev)   # a row per step: no step's statuses are erased
    torch.cuda.synchronize()
    for it in range(1 + steps):     # like the main loop: the timed steps are only enqueued, one synchronisation at the end
        if it == 1:
            circ.sync()
            torch.cuda.synchronize()
            t = time.perf_counter()
        circ.prove_dev(d_in.data_ptr(), B, d_seeds.data_ptr(), d_proofs.data_ptr(), pw, d_status[it].data_ptr())
    circ.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ok = bool((d_status.cpu().numpy() == 0).all())
    _p, _s, tm = circ.prove(inp, seeds=[0], timings=True)
    acc = verify(circ, d_proofs[B - 1].cpu().numpy().view(np.uint64), dg, cap)
    out = {"workload": f"batch of {B} plonky3-verifier proofs, inner Fibonacci trace 2^20 rows (outer n = 2^{int(info.degree_bits)} "
                       f"rows x {int(info.num_wires)} wires, LDE 2^{int(info.degree_bits) + 3})",
           "proofs_per_s": round(B * steps / dt, 3), "ms_per_step": round(dt / steps * 1e3, 1), "steps": steps,
           "all_statuses_ok": ok, "oracle_verifier_accepts": bool(acc),
           "single_proof_latency_ms": round(tm.as_dict()["total_ms"], 2), "circuit_build_s": round(build_s, 2),
           "native_p3_prover_s_two_proofs": round(p3_s, 2)}
    # VALU view of this configuration, from a PMC pass collected for exactly the current kernel sources
    # (tools/pmc_config5.sh writes profiles/*_config5_pmc_SQ_INSTS_VALU.json with the sources' hash)
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "plonky2.5_amd", "csrc")
    hh = hashlib.sha256()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".inc")) and fn != "capi.hip":   # capi.hip: host code only (the C ABI), no kernel
            hh.update(open(os.path.join(csrc, fn), "rb").read())
    sha = hh.hexdigest()[:16]
    out["valu"] = {"stale": True, "note": f"no profiles/*_config5_pmc_SQ_INSTS_VALU.json for the current kernel sources ({sha})"}
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_config5_pmc_SQ_INSTS_VALU.json")), reverse=True):
        try:
            pk = json.load(open(cand))
        except Exception:
            continue
        if pk.get("_meta", {}).get("csrc_sha") == sha:
            instr = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in pk.items() if k != "_meta")
            hz, hz_note = measured_clock(p25, torch, dev)
            ach = instr * out["proofs_per_s"]
            out["valu"] = {"wave_instr_per_proof": instr, "achieved_wave_instr_per_s": ach, "source": os.path.basename(cand),
                           "shader_clock_hz": hz, "shader_clock_source": hz_note, "frac_of_quarter_rate_4cyc": ach / (N_SIMD * hz / 4),
                           "per_row_vs_config3": "config 3 issues 3.97 G per proof of 2^16 rows; this circuit has 2^19"}
            break
    circ.close()
    return out


def main():
    args = parse_args()
    if args.no_cpu_baseline:
        args.cpu_baseline = "none"
    launcher = args.gpus > 1 and "RANK" not in os.environ
    if args.dist_backend == "nccl":
        n_vis = visible_gpus(in_child=launcher)
        if n_vis < args.gpus:
            if int(os.environ.get("RANK", "0")) == 0:
                fail(f"--gpus {args.gpus} but only {n_vis} GPU(s) are visible")
            sys.exit(2)
    if launcher:
        sys.exit(spawn_ranks(args))

    # libp25 sets this itself when it is loaded; torch may initialise HIP first, so set it here too
    # (hardware queues the runtime spreads the prover's streams over -- see capi.hip).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    if args.force_dist and args.gpus == 1 and "RANK" not in os.environ:
        # a world of one, no launcher: the rendezvous variables torch.distributed's env:// store reads
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.dist_backend == "nccl" else 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1 or args.force_dist
    if local_rank >= torch.cuda.device_count():
        # cannot join the process group without a device: say so (every rank prints its own line) and leave
        fail(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The collective timeout bounds what ranks != 0 may wait in the final barrier while rank 0 checks the run with the
        # oracle and assembles the JSON line (`rank0_post_region_s` in the record: tens of seconds): set explicitly, well
        # above that, instead of inheriting the backend's default watchdog.
        import datetime
        pg_timeout = datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S)
        if args.dist_backend == "nccl" and not args.dist_native:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        elif args.dist_native:
            # the library's own communicator carries the data path; torch.distributed is only the launcher's out-of-band channel
            # here (unique id, barriers, timing reductions, the input broadcast) and runs on gloo / host tensors -- a SECOND RCCL
            # communicator in the process would add its streams to the 24 hardware queues the proving streams live on
            # (profiles/r06_z_dist_native_vs_torch.txt: 124.5 instead of 145.4 proofs/s with both alive)
            dist.init_process_group(backend="gloo", timeout=pg_timeout)
        else:
            dist.init_process_group(backend="gloo", timeout=pg_timeout)

    p25 = ge.load_package()
    p25.device_init(local_rank)
    from plonky25_amd import dist as pdist
    from plonky25_amd import aggregate as pagg
    dev = torch.device("cuda", local_rank)
    cdev = dev if (args.dist_backend == "nccl" and not args.dist_native) else torch.device("cpu")   # where torch's collectives' tensors live
    host_threads = max(1, (os.cpu_count() or 1) // world)                 # host-side helpers: share the cores
    # Per-proof inputs: plonky3 proofs of the Fibonacci AIR.  Generated ONCE, on rank 0 (item 0 for log_n = 6 is the
    # reference's artifact through the library's own reader; further valid proofs of the same statement -- other
    # PoW witnesses, hence other query indices -- come from the native plonky3 prover), then broadcast.
    n_var = max(1, min(args.distinct, args.total if args.total else args.batch))
    if args.log_n == 6:
        with open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")) as f:
            inputs, p3cfg = p25.p3_proof_from_json(f.read())
    elif rank == 0:  # BASELINE config 5 and friends: inner STARK with 2^log_n rows
        inputs, p3cfg = p25.p3_prove_fibonacci(args.log_n, 100, 16, threads=host_threads)
    else:
        inputs, p3cfg = None, None
    if rank == 0:
        variants = [inputs] + [p25.p3_prove_fibonacci(args.log_n, 100, 16, pow_start=v << 24, threads=host_threads)[0]
                               for v in range(1, n_var)]
        var_arr = np.stack(variants)
    if distributed:
        if args.log_n != 6:   # the shape of a 2^log_n-row proof: 9 int32 fields of p25_p3_config + the input length
            meta = np.zeros(10, dtype=np.int64)
            if rank == 0:
                meta[:9] = [getattr(p3cfg, n) for n, _ in p3cfg._fields_]
                meta[9] = var_arr.shape[1]
            meta = pdist.broadcast_int64(meta if rank == 0 else None, (10,), cdev).view(np.int64)
            if rank != 0:
                p3cfg = p25.P3Config(*[int(x) for x in meta[:9]])
            ni_b = int(meta[9])
        else:
            ni_b = int(inputs.size)
        var_arr = pdist.broadcast_int64(var_arr if rank == 0 else None, (n_var, ni_b), cdev)
        variants = [var_arr[i] for i in range(n_var)]
        inputs = variants[0]

    # circuit: built once per shape by the host code, tables made resident on the GPU
    t0 = time.time()
    circuit = p25.Circuit.build_p3_verifier(p3cfg)
    info = circuit.info
    digest, cs_cap = circuit.digest()  # forces the device-side tables (constants/sigmas commitment)
    build_s = time.time() - t0
    # weak scaling (default): --batch proofs per GPU per step.  strong scaling (--total): that many per step in all.
    if args.total:
        g_start, g_stop = pdist.shard_range(args.total, rank, world)
        B, n_step_total = g_stop - g_start, args.total
    else:
        B, n_step_total = args.batch, args.batch * world
        g_start = rank * B
    pw = int(info.proof_words)
    host_in = np.stack([variants[(g_start + i) % len(variants)] for i in range(max(B, 1))])[:B]
    host_seeds = np.arange(B, dtype=np.uint64) + np.uint64(g_start)                          # distinct filler seeds
    d_inputs = torch.from_numpy(host_in.view(np.int64)).to(dev)                            # [B][ni]
    d_seeds = torch.from_numpy(host_seeds.view(np.int64)).to(dev)
    # Step k writes proofs into buffer k & 1 and its statuses into row k of a per-step status array: the steps are
    # pipelined (the library orders step k+1's proof i behind step k's proof i on the same stream, nothing else), so no
    # step's statuses are erased before they are inspected, and -- with N > 1 -- step k's finished proofs are gathered
    # on a side stream underneath step k+1's proving instead of draining the 16 proving streams at every step.
    n_steps_all = args.warmup + args.steps
    d_proofs = [torch.zeros((max(B, 1), pw), dtype=torch.int64, device=dev)[:B] for _ in range(2)]
    d_status_all = torch.zeros((max(n_steps_all, 1), max(B, 1)), dtype=torch.int32, device=dev)[:, :B]
    native = distributed and args.dist_native
    if args.dist_native and args.dist_backend != "nccl":
        fail("--dist-native is the RCCL path: it needs --dist-backend nccl")
    gatherer = pdist.ProofGatherer(n_step_total, pw, cdev, slots=2) if (distributed and not native) else None   # buffers allocated once
    comm, n_all, n_st, counts = None, None, None, None
    if native:
        # the launcher's out-of-band channel hands rank 0's unique id to every rank; everything on the data path is the library's
        uid = torch.zeros(128, dtype=torch.uint8, device=cdev)
        if rank == 0:
            uid = torch.frombuffer(bytearray(p25.comm_unique_id()), dtype=torch.uint8).to(cdev)
        dist.broadcast(uid, src=0)
        comm = p25.Comm(bytes(uid.cpu().numpy().tobytes()), rank, world)
        counts = pdist.shard_sizes(n_step_total, world) if args.total else [B] * world
        if rank == 0:      # receive buffers, one per pipelined slot, allocated once
            n_all = [torch.zeros((n_step_total, pw), dtype=torch.int64, device=dev) for _ in range(2)]
            n_st = [torch.full((n_step_total,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
    # device memory: the library adapts the number of proofs in flight to what is free, which would silently change the
    # schedule being measured -- refuse instead (one JSON error line, all ranks exit 2)
    nW, NCh, NPp = int(info.num_wires), int(info.num_challenges), int(info.num_partial_products)
    n_rows, n_lde = 1 << int(info.degree_bits), 1 << (int(info.degree_bits) + 3)
    nzc, nqc = NCh * (1 + NPp), NCh * int(info.quotient_degree_factor)
    ctx_est = 8 * (3 * nW * n_rows + nW * n_lde + 2 * nzc * n_rows + nzc * n_lde + 3 * NCh * n_lde + nqc * n_lde
                   + 3 * p25.merkle_tree_words(n_lde, 4) + 16 * n_rows + 6 * n_lde)
    free_b, total_b = torch.cuda.mem_get_info()
    in_flight_fit = int(max(0, free_b - total_b // 20) // ctx_est)
    want_in_flight = min(16, max(B, 1))
    short = 1 if (args.log_n == 6 and in_flight_fit < want_in_flight) else 0
    if distributed:
        sh_t = torch.tensor([short], dtype=torch.int32, device=cdev)
        dist.all_reduce(sh_t, op=dist.ReduceOp.MAX)
        any_short = int(sh_t.item())
    else:
        any_short = short
    if any_short:
        if short:
            sys.stderr.write(f"rank {rank}: {free_b / 1e9:.1f} GB free of {total_b / 1e9:.1f} GB holds {in_flight_fit} proof "
                             f"contexts of {ctx_est / 1e9:.2f} GB, {want_in_flight} needed\n")
        if rank == 0:
            fail("a rank's GPU has too little free memory for the proofs-in-flight schedule (details on stderr)")
        sys.exit(2)
    nccl = distributed and args.dist_backend == "nccl"
    # ONE side stream for the gathers of both buffers (side[0] is side[1]): every stream that has carried work keeps its
    # hardware queue, the library's pool (16) + one main stream per circuit + torch's + RCCL's come close to the 24 there
    # are, and streams sharing a queue run in order (profiles/r04_stream_pool.txt: the distributed branch lost 2.5 % of
    # the pipelined tree with two).  Gather k-1 is issued after step k, so at step k the stream holds gather k-2 at most.
    side = [torch.cuda.Stream(device=dev)] * 2 if (distributed and not native) else None
    if native:      # the communicator's own stream, seen through torch only to time the gathers with events
        side = [torch.cuda.ExternalStream(comm.stream, device=dev)] * 2
    g_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps_all)] if nccl else None
    torch.cuda.synchronize()  # inputs are resident in HBM before anything is timed

    gather_host_s = [0.0]
    gathered = [None, None]
    pending = []       # gloo test mode: (step, event) whose host-side gather has not run yet
    step_no = [0]
    gather_issued = [0]      # steps whose gather has been issued (it lags the step by one)

    def gather_on_host(k, ev):   # test mode only (every rank on GPU 0, collectives on host copies)
        ev.synchronize()
        g0 = time.perf_counter()
        gathered[0], gathered[1] = gatherer.gather(d_proofs[k & 1].cpu(), d_status_all[k].cpu(), slot=k & 1)
        gather_host_s[0] += time.perf_counter() - g0

    GATHER_MARK = pagg.TREE_MARK_SLOTS     # the top two mark slots: the gather's (the others are the pipelined tree's, aggregate.DeviceTree)

    def issue_gather(k):
        """The final aggregation step of batch k: finished proofs gathered onto rank 0 (RCCL over xGMI) on a side stream
        that waits -- on the device -- for the mark taken when step k had been enqueued."""
        buf = k & 1
        if native:     # p25_gather_proofs: waits for the mark on the device, grouped ncclSend / ncclRecv, all on the library's stream
            circuit.stream_wait_mark(GATHER_MARK + buf, comm.stream)    # (also inside the call; here so that the events time the gather alone)
            g_ev[k][0].record(side[buf])
            comm.gather(circuit, GATHER_MARK + buf, d_proofs[buf].data_ptr(), pw, d_status_all[k].data_ptr(), counts, 0,
                        n_all[buf].data_ptr() if rank == 0 else None, n_st[buf].data_ptr() if rank == 0 else None)
            g_ev[k][1].record(side[buf])
            if rank == 0:
                offs = [sum(counts[:q]) for q in range(world + 1)]
                gathered[0] = [n_all[buf][offs[q]:offs[q + 1]] for q in range(world)]
                gathered[1] = [n_st[buf][offs[q]:offs[q + 1]] for q in range(world)]
            return
        with torch.cuda.stream(side[buf]):
            circuit.stream_wait_mark(GATHER_MARK + buf, side[buf].cuda_stream)
            if nccl:
                g_ev[k][0].record(side[buf])
                gathered[0], gathered[1] = gatherer.gather(d_proofs[buf], d_status_all[k], slot=buf)
                g_ev[k][1].record(side[buf])
            else:
                ev = torch.cuda.Event()
                ev.record(side[buf])
                pending.append((k, ev))

    def step():
        k = step_no[0]
        step_no[0] += 1
        buf = k & 1
        if nccl and k >= 2:   # gather k-2 (the only work side[buf] holds) still reads the buffer this step overwrites
            circuit.wait_stream(side[buf].cuda_stream)
        while pending and pending[0][0] <= k - 2:   # test mode (host copies): the same protection, on the host
            gather_on_host(*pending.pop(0))
        if B:
            circuit.prove_dev(d_inputs.data_ptr(), B, d_seeds.data_ptr(), d_proofs[buf].data_ptr(), pw,
                              d_status_all[k].data_ptr())
        if not distributed:
            return
        # The gather of step k is issued ONE STEP LATE, after step k+1 has been enqueued: streams share hardware queues,
        # and a wait that is not yet satisfied when it reaches the head of its queue would hold up the next step's
        # kernels enqueued behind it; the host never blocks either way.
        circuit.mark(GATHER_MARK + buf)
        while gather_issued[0] < k:
            issue_gather(gather_issued[0])
            gather_issued[0] += 1

    def drain():
        while distributed and gather_issued[0] < step_no[0]:   # the last step's gather
            issue_gather(gather_issued[0])
            gather_issued[0] += 1
        while pending:
            gather_on_host(*pending.pop(0))
        circuit.sync()
        if native:
            comm.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    gather_host_s[0] = 0.0
    circuit.kernel_stats(enable=True, reset=True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    local_elapsed = time.perf_counter() - t0
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t_region_end = time.perf_counter()
    per_rank = None
    if distributed:
        gather_s = (sum(e0.elapsed_time(e1) for e0, e1 in g_ev[args.warmup:]) * 1e-3) if nccl else gather_host_s[0]
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([local_elapsed, gather_s, float(B)], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "proofs_per_step": int(x[2]), "proofs_per_s": round(float(x[2]) * args.steps / float(x[0]), 2),
                     "gather_ms_per_step": round(float(x[1]) / max(1, args.steps) * 1e3, 3)} for r, x in enumerate(allr)]
    k_ms_busy, k_launches_busy = circuit.kernel_stats(enable=False, reset=True)

    # statuses of EVERY timed step (a failure in an early step must not be counted as proofs)
    statuses = d_status_all[args.warmup:].cpu().numpy()
    ok = bool((statuses == 0).all())
    if distributed:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    last = d_proofs[(n_steps_all - 1) & 1] if n_steps_all else d_proofs[0]

    # --- the batch folded to ONE proof, sharded like the batch (outside the timed region; all ranks take part) ------
    # Every rank folds the first n_agg proofs of ITS last step to one root on its own GPU, the N roots (not the leaves)
    # are gathered over RCCL, rank 0 proves one N-to-1 aggregate on top.  Collectives only where every rank reaches them:
    # a local failure is agreed on first.
    agg_state = None
    if args.aggregate != 0 and args.aggregate_arity == 0 and ok:
        args.aggregate_arity = pagg.widest_arity(circuit)      # deterministic: the same on every rank
    n_agg = min(args.aggregate if args.aggregate >= 0 else 64, B)
    if distributed:
        na_t = torch.tensor([n_agg], dtype=torch.int32, device=cdev)
        dist.all_reduce(na_t, op=dist.ReduceOp.MIN)     # every shard folds the same shape
        n_agg = int(na_t.item())
    if n_agg * world >= 2 and ok:
        local_leaves = last[:n_agg].cpu().numpy().view(np.uint64)
        agg_state = pagg.fold_sharded(circuit, local_leaves, args.aggregate_arity, cdev, distributed)
        # ... and what a level-1 aggregate proof costs the machine with enough of them in flight (rank 0 only, no collective)
        f0 = agg_state.get("fold")
        if rank == 0 and f0 and f0["owned"] and n_agg >= f0["levels"][0]["arity"]:
            try:
                k1 = f0["levels"][0]["arity"]
                rate1 = pagg.circuit_throughput(f0["owned"][0], np.concatenate([local_leaves[i] for i in range(k1)]), dev,
                                                count=128, steps=3)   # 384 timed proofs: the ramp and the drain of 16 pipelines are ~1 % of that
                agg_state["level1_throughput"] = {"aggregate_proofs_per_s": round(rate1, 2), "ms_per_aggregate_proof": round(1e3 / rate1, 3),
                                                  "batch": 128, "steps": 3,
                                                  "note": "the level-1 aggregation circuit by itself, 16 proofs in flight, steps enqueued back to back"}
            except Exception as e:
                agg_state["level1_throughput"] = {"error": str(e)[:200]}
        pagg.release_fold(agg_state)     # the checker's data kept, the fold's circuits (tens of GB of working sets) closed

    # --- the same tree PIPELINED: device-resident, enqueue-only, lagged one step per level -----------------------------
    # What a production batch prover runs: every step = B leaf proofs + one instance of every level of the aggregation
    # tree (B -> B/8 -> ... -> 1 per rank, over the leaves of earlier steps), nothing synchronised until the end.
    # Measures leaf proofs/s INCLUDING their aggregation directly, in steady state.
    pipe = None
    nl = B
    pipe_go = args.aggregate != 0 and ok and nl >= 2 and (agg_state is None or not agg_state.get("error"))
    if distributed:      # the block holds collectives: every rank enters it or none does (shards may differ in size)
        pg_t = torch.tensor([1 if pipe_go else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(pg_t, op=dist.ReduceOp.MIN)
        pipe_go = bool(pg_t.item())
    if pipe_go:
        perr, tree = None, None
        K_pipe = 3
        try:
            tree = pagg.DeviceTree(circuit, nl, args.aggregate_arity, dev, leaf_batch=B)
            d_status_pipe = torch.zeros((len(tree.levels) + 1 + K_pipe, B), dtype=torch.int32, device=dev)

            def pleaves(buf, j):
                circuit.prove_dev(d_inputs.data_ptr(), B, d_seeds.data_ptr(), buf.data_ptr(), pw, d_status_pipe[j].data_ptr())

            for _ in range(len(tree.levels) + 1):      # fill the pipeline: every level has run once (contexts, tables)
                tree.step(pleaves)
            tree.sync(); torch.cuda.synchronize()
        except Exception as e:
            perr = str(e)[:300]
        fine = perr is None
        if distributed:
            fl = torch.tensor([1 if fine else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(fl, op=dist.ReduceOp.MIN)
            fine = bool(fl.item())
            if fine:
                dist.barrier()
        if fine:
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for _ in range(K_pipe):
                tree.step(pleaves)
            tree.sync(); torch.cuda.synchronize()
            p_elapsed = time.perf_counter() - tp0
            if distributed:
                p_elapsed = pdist.max_over_ranks(p_elapsed, cdev)
            tree.flush(); tree.sync(); torch.cuda.synchronize()       # the levels still owed to the last steps, untimed
            last_step = tree.leaf_steps - 1
            root, root_ok = tree.root(last_step)
            leaves_ok = bool((d_status_pipe.cpu().numpy() == 0).all())
            pipe = {"tree": tree, "elapsed": p_elapsed, "steps": K_pipe, "leaves": nl, "root": root,
                    "statuses_ok": bool(root_ok and leaves_ok),
                    "caps": tree.leaf_proofs(last_step)[:nl, :pagg.CAP_WORDS].cpu().numpy().view(np.uint64).copy()}
        else:
            pipe = {"error": perr or "another rank's pipelined tree failed"}

    if rank == 0:
        # --- outside the timed region ---------------------------------------------------------------
        # the dominant kernel with the GPU to itself: single-proof passes (one stream), HIP events around it
        circuit.kernel_stats(enable=True, reset=True)
        alone = 8
        tm = None
        gpu_proof0 = None
        for i in range(alone):
            _p, _s, tm = circuit.prove(variants[i % len(variants)], seeds=[i], timings=True)
            if i == 0:
                gpu_proof0 = _p[0].copy()   # inputs = item 0, filler seed 0: the proof the one-thread CPU leg reproduces
        k_ms_alone, k_launches_alone = circuit.kernel_stats(enable=False, reset=True)
        # correctness: the oracle verifier accepts proofs spread over the last batch
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_binding import Oracle           # the checker: verification and cpu_baseline only
        ora = Oracle()
        oc = ora.load_circuit(circuit.to_blob())
        # ... spread over the WHOLE step: with N > 1 the proofs checked are the gathered ones, every rank's block
        if distributed:
            all_blocks = [b.cpu().numpy().view(np.uint64) for b in gathered[0]]
            all_p = np.concatenate(all_blocks) if all_blocks else np.zeros((0, pw), dtype=np.uint64)
            gathered_ok = bool(all(int((g.cpu() != 0).sum()) == 0 for g in gathered[1])) and all_p.shape[0] == n_step_total
        else:
            all_p, gathered_ok = last.cpu().numpy().view(np.uint64), True
        nver = max(1, min(args.verify, all_p.shape[0]))
        ver_idx = sorted({int(round(k * (all_p.shape[0] - 1) / max(1, nver - 1))) for k in range(nver)})
        ver_fail = []
        for i in ver_idx:
            code, msg = oc.verify(all_p[i], digest, cs_cap)
            if code != 0:
                ver_fail.append((i, msg))
        n_big = 1 << (int(info.degree_bits) + 3)
        algo_bytes = n_big * int(info.num_wires) * 8 + n_big * 32   # read the LDE once, write one digest per leaf
        ms_alone = k_ms_alone / max(1, k_launches_alone)
        ms_busy = k_ms_busy / max(1, k_launches_busy)
        achieved = algo_bytes / (ms_alone * 1e-3) / 1e9 if ms_alone > 0 else 0.0
        ms_all = (k_ms_alone + k_ms_busy) / max(1, k_launches_alone + k_launches_busy)
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
        traffic, traffic_note = None, "no PMC file"
        tp = os.path.join(ROOT, "profiles", "pmc_hash_leaves.json")
        if os.path.exists(tp):
            try:
                import hashlib
                tj = json.load(open(tp))
                hh = hashlib.sha256()
                for fn in ("kernels_hash.hip", "poseidon.h", "poseidon_p3r.h", "gl.h"):
                    hh.update(open(os.path.join(ROOT, "plonky2.5_amd", "csrc", fn), "rb").read())
                if tj.get("kernel_source_sha") == hh.hexdigest()[:16]:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/ (collected at "
                                    f"{tj.get('head', '?')}; the kernel's sources are unchanged since)")
                else:
                    traffic_note = "profiles/pmc_hash_leaves.json was collected for other kernel sources: re-run the PMC passes"
            except Exception:
                traffic = None
        total_proofs = n_step_total * args.steps
        # Integer-VALU view of the same run (the bound that actually binds): wave-level VALU instructions
        # per proof from the latest committed PMC pass (SQ_INSTS_VALU, profiles/*_pmc_SQ_INSTS_VALU.json,
        # written by tools/collect_profiles.sh) x proofs/s per GPU, against two ceilings.
        valu = None
        import glob
        import hashlib
        csrc = os.path.join(ROOT, "plonky2.5_amd", "csrc")
        hh = hashlib.sha256()
        for fn in sorted(os.listdir(csrc)):
            if fn.endswith((".hip", ".h", ".inc")) and fn != "capi.hip":   # capi.hip: host code only (the C ABI), no kernel
                hh.update(open(os.path.join(csrc, fn), "rb").read())
        csrc_sha = hh.hexdigest()[:16]
        clock_hz, clock_note = measured_clock(p25, torch, dev)
        # only a PMC pass collected for exactly these kernel sources counts (tools/pmc_summary.py writes _meta.csrc_sha)
        vp, per_kernel = "", None
        for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_SQ_INSTS_VALU.json")), reverse=True):
            try:
                pk = json.load(open(cand))
            except Exception:
                continue
            if pk.get("_meta", {}).get("csrc_sha") == csrc_sha:
                vp, per_kernel = cand, {k: v for k, v in pk.items() if k != "_meta"}
                break
        if per_kernel is None:
            valu = {"stale": True, "note": "no profiles/*_pmc_SQ_INSTS_VALU.json was collected for the current kernel sources "
                                           f"(csrc sha {csrc_sha}): re-run tools/collect_profiles.sh", "shader_clock_hz": clock_hz,
                    "shader_clock_source": clock_note}
        elif args.log_n == 6:
            instr_per_proof = sum(v.get("SQ_INSTS_VALU", 0.0) for v in per_kernel.values())
            ach = instr_per_proof * (total_proofs / elapsed) / world
            peak4, peak2 = N_SIMD * clock_hz / 4, N_SIMD * clock_hz / 2
            valu = {"wave_instr_per_proof": instr_per_proof, "achieved_wave_instr_per_s": ach,
                    "source": os.path.basename(vp), "csrc_sha": csrc_sha,
                    "shader_clock_hz": clock_hz, "shader_clock_source": clock_note,
                    "frac_of_plain_issue_2cyc": ach / peak2, "frac_of_quarter_rate_4cyc": ach / peak4,
                    "frac_of_measured_mix_ceiling": ach / (peak4 * (4.0 / 4.13)),
                    "leaf_hash_share": sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in per_kernel.items()
                                           if "k_hash_leaves" in k) / instr_per_proof,
                    "note": "ceilings: 1 wave-instruction / SIMD / 2 cycles is the guide's plain-VALU issue rate; every "
                            "VOP3 / carry / v_mad_u64_u32 instruction (the whole mix here) is of the 4-cycle class; the "
                            "measured ceiling is what the hash kernels reach with 16 streams of them in flight and nothing "
                            "else: 4.13 cycles per instruction (profiles/r03_pipeline_model_experiments.txt; a single "
                            "launch, with its ragged last wave round, measures 4.6).  The proving pipeline stays below it "
                            "because k_ntt_tile and k_quotient (57 % / 55 % VALU-busy alone) time-share the chip with the "
                            "hash kernels rather than fill their issue slots"}
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
            "scaling": "strong" if args.total else "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, p = 2^64 - 2^32 + 1)",
            "data": f"synthetic: {len(variants)} distinct plonky3 proofs of fibonacci(2^{args.log_n}) "
                    "(item 0 = the reference's artifacts/proof_fibonacci.json for log_n 6; the others from the native "
                    "p3 prover with other PoW witnesses) cycled through the batch, distinct filler seeds",
            "config": {"workload": (f"batch of {n_step_total} independent " if args.total else f"batch of {B} independent ")
                                   + f"{'fib-64' if args.log_n == 6 else f'fibonacci(2^{args.log_n})'} plonky3-verifier proofs "
                                   + ("sharded over the GPUs " if args.total else "per GPU ") +
                                   f"(n = 2^{int(info.degree_bits)} rows x 135 wires, LDE 2^{int(info.degree_bits) + 3}), "
                                   f"{world} GPU(s), replicas + RCCL gather"
                                   + (" (p25_gather_proofs: librccl behind the C ABI)" if native else "")
                                   + ("" if args.dist_backend == "nccl" else " [TEST MODE: all ranks on GPU 0, gloo]"),
                       "proofs_per_gpu_per_step": B, "proofs_per_step_total": n_step_total,
                       "all_statuses_ok": ok, "gathered_complete_and_ok": gathered_ok,
                       "oracle_verifier_accepts": not ver_fail, "oracle_verified_indices": ver_idx,
                       "circuit_build_s": round(build_s, 2), "head": head, "runtime": p25.runtime_info().as_dict(),
                       "single_proof_latency_ms": round(tmd["total_ms"], 3),
                       "phase_ms_single_proof": {k: round(v, 3) for k, v in tmd.items()}},
            "roofline": {"bound": "valu", "contract_view": "hbm",   # achieved / peak / unit / frac / traffic below: the contract's HBM roof
                         "view": "achieved / peak / frac are the HBM view the contract prescribes (algorithmic bytes of the dominant "
                                 "kernel per launch / its launch duration against 8 TB/s); the roof that BINDS is integer-VALU issue: "
                                 "`valu` below (wave-instructions per second against 1 per SIMD per 4 cycles)",
                         "kernel": "k_hash_leaves_wide (Poseidon sponge, 2^19 leaves x 135 words)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "avg_launch_ms": ms_alone, "launches": int(k_launches_alone), "algorithmic_bytes": algo_bytes,
                         "avg_launch_ms_timed_region": ms_busy, "launches_timed_region": int(k_launches_busy),
                         # the figure a rocprofv3 --stats AverageNs over the WHOLE run gives: every launch, overlapped or not
                         "avg_launch_ms_all_launches": ms_all,
                         "frac_all_launches": (algo_bytes / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_all > 0 else 0.0,
                         "note": "integer-VALU bound (17 Poseidon permutations per 1,080-B leaf), not HBM bound. achieved/frac "
                                 "use avg_launch_ms = the kernel with the GPU to itself (single-proof passes after the timed "
                                 "region, HIP events on the proving stream; this is what rocprofv3's per-kernel duration "
                                 "shows for non-overlapped launches).  avg_launch_ms_timed_region is the same bracket with "
                                 "up to 16 proofs in flight: co-resident kernels time-slice the SIMDs, so it measures "
                                 "residency, not speed",
                         "valu": valu, "hbm_phases": hbm_phases, "hbm_overall": hbm_overall},
        }
        if per_rank:
            out["per_rank"] = per_rank
        if ver_fail:
            out["config"]["oracle_verifier_failures"] = [f"{i}: {m}" for i, m in ver_fail]
        def verify_with_oracle(circ, proof, dg=None, cap=None):
            o2 = ora.load_circuit(circ.to_blob())
            if dg is None:
                dg, cap = circ.digest()
            return o2.verify(proof, dg, cap)[0] == 0

        # --- the batch folded to ONE proof: rank 0 checks the root and reports ---------------------------------------
        if agg_state is not None:
            if agg_state.get("error"):
                out["aggregation"] = {"error": agg_state["error"]}
            else:
                try:
                    ck = agg_state["checker"]
                    root, root_pis = ck["root"], ck["root_public_inputs"]
                    want = pagg.expected_commitment(list(agg_state["caps"]), args.aggregate_arity, ora.hash_no_pad, n_shards=world)
                    leaves = agg_state["leaves_per_rank"] * world
                    tree_total = agg_state["tree_s_max"] + agg_state["roots_gather_ms"] * 1e-3 + ck["final_tree_s"]
                    leaf_s = leaves / (total_proofs / elapsed)   # the leaves at the measured whole-job rate
                    root_ok = ora.load_circuit(ck["top_blob"]).verify(root, ck["digest"], ck["cap"])[0] == 0
                    out["aggregation"] = {
                        "leaves": leaves, "leaves_per_rank": agg_state["leaves_per_rank"], "ranks": world,
                        "max_children_per_aggregation_circuit": args.aggregate_arity,
                        "levels": ck["levels"], "level1_throughput": agg_state.get("level1_throughput"),
                        "root_public_inputs": root_pis,
                        "root_public_inputs_commit_to_the_leaves": root_pis == want,
                        "shard_tree_prove_s_max_over_ranks": round(agg_state["tree_s_max"], 4),
                        "roots_gather_ms": round(agg_state["roots_gather_ms"], 3),
                        "cross_rank_prove_s": round(ck["final_tree_s"], 4), "tree_prove_s": round(tree_total, 4),
                        "tree_circuit_build_s_once_per_shape": round(ck["build_s"], 2),
                        "root_proof_words": int(root.size), "oracle_verifier_accepts_root": bool(root_ok),
                        "leaf_prove_s_at_measured_rate": round(leaf_s, 4),
                        "leaf_equivalent_proofs_per_s_including_aggregation": round(leaves / (leaf_s + tree_total), 2),
                        "note": "every rank folds the first proofs of its last timed step to one root on its own GPU (levels "
                                "proved as batches, the shard trees run concurrently); the N roots -- not the leaves -- are "
                                "gathered over RCCL and rank 0 proves one N-to-1 aggregate on top; every level is an aggregation "
                                "circuit (recursive verifier of its children + 4 public inputs committing to them); circuit "
                                "builds are once per shape and excluded like the reference's build()"}
                except Exception as e:  # never lose the headline line to the optional block
                    out["aggregation"] = {"error": str(e)[:300]}
        if pipe is not None and isinstance(out.get("aggregation"), dict):
            if pipe.get("error"):
                out["aggregation"]["pipelined"] = {"error": pipe["error"]}
            else:
                try:
                    tree = pipe["tree"]
                    got = [int(v) for v in tree.top.public_inputs(pipe["root"])]
                    want = pagg.expected_commitment(list(pipe["caps"]), args.aggregate_arity, ora.hash_no_pad)
                    per_step = pipe["leaves"] + tree.aggregates_per_step
                    rate = world * B * pipe["steps"] / pipe["elapsed"]
                    out["aggregation"]["pipelined"] = {
                        "leaf_equivalent_proofs_per_s": round(rate, 2), "fraction_of_unaggregated_rate": round(rate / (total_proofs / elapsed), 4),
                        "steps": pipe["steps"], "ms_per_step": round(pipe["elapsed"] / pipe["steps"] * 1e3, 1),
                        # what an aggregate proof costs the machine: the step's extra time over its leaves at the un-aggregated rate
                        # (None when the un-aggregated steps were slower than the pipelined ones: the gloo TEST MODE, whose gather blocks the host)
                        "machine_ms_per_aggregate_proof": (lambda x: round(x, 2) if x > 0 else None)(
                            (pipe["elapsed"] / pipe["steps"] - B / (total_proofs / elapsed / world)) * 1e3 / max(1, tree.aggregates_per_step)),
                        "leaf_proofs_per_rank_per_step": B, "leaves_folded_per_rank_per_step": pipe["leaves"],
                        "aggregate_proofs_per_rank_per_step": tree.aggregates_per_step, "proofs_of_any_kind_per_rank_per_step": B - pipe["leaves"] + per_step,
                        "levels": [{"arity": L["k"], "children": L["children"], "proofs": L["n"], "circuit_rows_log2": int(L["circ"].info.degree_bits)}
                                   for L in tree.levels],
                        "all_statuses_ok": pipe["statuses_ok"], "root_public_inputs_commit_to_the_leaves": got == want,
                        "oracle_verifier_accepts_root": bool(verify_with_oracle(tree.top, pipe["root"])),
                        "tree_circuit_build_s_once_per_shape": round(tree.build_s, 2),
                        "note": "steady state: every timed step = one leaf batch + one instance of every level of the aggregation tree "
                                "(over the leaves of earlier steps: level l lags l steps), device-resident (a level proves straight "
                                "on the buffer the level below writes: an aggregator's inputs are its children's flat proofs back "
                                "to back), enqueue-only, ordered by events (p25_circuit_mark / p25_circuit_wait_mark); rank 0's "
                                "root of the last step checked after the pipeline has been flushed"}
                    tree.close()
                except Exception as e:
                    out["aggregation"]["pipelined"] = {"error": str(e)[:300]}
        # --- the other single-GPU BASELINE configs in the same record ----------------------------------------------
        if args.extra_configs == "auto" and world == 1 and args.log_n == 6 and not args.total:
            cfgs = {"config2_single_proof": {"workload": "one fib-64 verifier proof alone on the GPU (latency-oriented kernel forms)",
                                             "latency_ms": round(tmd["total_ms"], 3),
                                             "phase_ms": {k: round(v, 3) for k, v in tmd.items()}},
                    "config3_batch256": {"proofs_per_s": round(total_proofs / elapsed, 3), "this_record": True}}
            try:
                circuit.close()   # free the fib-64 contexts (26 GB) before the 2^19-row circuit's (13 GB each)
                cfgs["config5_inner_2pow20"] = bench_config5(p25, np, torch, dev, host_threads,
                                                             lambda c, pr, dg, cap: verify_with_oracle(c, pr, dg, cap))
            except Exception as e:
                cfgs["config5_inner_2pow20"] = {"error": str(e)[:300]}
            out["configs"] = cfgs
        if args.cpu_baseline != "none" and world == 1:  # reported baseline: rank 0, N = 1 only
            oc.digest()  # constants/sigmas commitment is per-circuit, excluded like the reference's build()
            model = cpu_model()
            # (i) BASELINE config 1: the reference's prover is single-threaded (Cargo.toml:15-18 no `parallel`)
            pr1, st1, _per1, wall1 = oc.prove_many(inputs[None, :], np.array([0], dtype=np.uint64), threads=1,
                                                  want_proofs=True)
            bit_exact = bool(gpu_proof0 is not None and (pr1[0] == gpu_proof0).all())
            untuned = {"value": 1.0 / wall1, "unit": "proofs/s", "cores": 1, "kind": "port",
                       "sample": f"1 full fib-64 proof (witness generation + prove) by the oracle C++ restatement (scalar) on ONE "
                                 f"pinned thread: {wall1:.1f} s, status {int(st1[0])}",
                       "gpu_proof_bit_exact_vs_this_cpu_proof": bit_exact}
            # (i') the TUNED leg: the same prover with its Merkle hashing and its quotient evaluation eight
            # at a time on AVX-512 (oracle/ref_hash_x8.cpp, ref_quotient_x8.cpp, the FFT stages of ref_fft.cpp), same thread count, validated against the scalar
            # proof byte for byte.  This is the number reported as cpu_baseline.value; the scalar one stays beside it.
            cb, tuned_note = None, "no AVX-512 on this host"
            if ora.set_tuned(True):
                try:
                    prt, stt, _pert, wallt = oc.prove_many(inputs[None, :], np.array([0], dtype=np.uint64), threads=1,
                                                           want_proofs=True)
                    same = bool((prt[0] == pr1[0]).all())
                    cb = {"value": 1.0 / wallt, "unit": "proofs/s", "cores": 1, "kind": "port-tuned", "cpu": model,
                          "sample": f"1 full fib-64 proof by the oracle with AVX-512 Merkle hashing, quotient evaluation and FFT butterflies (8 lanes) "
                                    f"on ONE pinned thread: {wallt:.1f} s, status {int(stt[0])}; the scalar oracle: {wall1:.1f} s",
                          "bytes_equal_to_the_scalar_oracle_proof": same,
                          "gpu_proof_bit_exact_vs_this_cpu_proof": bool(gpu_proof0 is not None and (prt[0] == gpu_proof0).all()),
                          "untuned": untuned}
                    if not same or int(stt[0]) != 0:     # a tuned leg that disagrees with the checker is not a baseline -- and is said so
                        cb, tuned_note = None, f"REJECTED: status {int(stt[0])}, bytes equal to the scalar proof: {same}"
                except Exception as e:
                    cb, tuned_note = None, f"failed: {str(e)[:200]}"
            if cb is None:
                ora.set_tuned(False)
                cb = dict(untuned, cpu=model, tuned_leg=tuned_note)
            if "configs" in out:
                out["configs"]["config2_single_proof"]["bit_exact_vs_cpu_port"] = bit_exact
            if args.cpu_baseline == "full":
                # (ii) the whole host: G proofs in flight, each on T threads of the oracle's persistent pool,
                # G x T = the physical cores.  (One single-threaded proof per core was measured too: 128 working sets
                # of ~4 GB compete for the memory system and the host delivers 0.21 proofs/s -- docs/HISTORY.md section 4.)
                cores = physical_cores()
                if args.cpu_cores:
                    cores = cores[:args.cpu_cores]
                T = min(16, len(cores))
                G = max(1, len(cores) // T)
                os.sched_setaffinity(0, set(cores[:G * T]))
                many_in = np.stack([variants[i % len(variants)] for i in range(G)])
                _pr, stn, pern, walln = oc.prove_many(many_in, np.arange(G, dtype=np.uint64), threads=G,
                                                      want_proofs=False, threads_per_proof=T)
                cb["all_cores"] = {"value": G / walln, "unit": "proofs/s", "cores": G * T,
                                   "kind": cb["kind"],
                                   "sample": f"{G} independent fib-64 proofs in flight, {T} threads each (persistent pool) on "
                                             f"{G * T} physical cores: wall {walln:.1f} s, per-proof "
                                             f"{float(pern.min()):.1f}-{float(pern.max()):.1f} s, all ok: {bool((stn == 0).all())}"}
            ora.set_tuned(False)
            out["cpu_baseline"] = cb
        # what the other ranks sit out in the final barrier (everything rank 0 did alone since the timed region ended,
        # the collectives of the aggregation blocks included) against the process group's timeout
        out["rank0_post_region_s"] = round(time.perf_counter() - t_region_end, 2)
        out["collective_timeout_s"] = COLLECTIVE_TIMEOUT_S
        print(json.dumps(out), flush=True)
    if native:
        comm.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
