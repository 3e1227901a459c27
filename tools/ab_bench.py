#!/usr/bin/env python3
"""A/B of library builds on ONE box, back to back: the batch proving loop of bench.py (device-resident inputs, steps
only enqueued, one sync at the end) for each library given, each in its own child process, `--rounds` times
interleaved.  Prints proofs/s per library and round, and the single-proof phase times of each.
usage: ab_bench.py [--batch 256] [--steps 3] [--rounds 2] [--agg N] [--pipe K] name=path/to/libp25_x.so ...   (name "base" = the product)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child_config5(lib, steps):
    """BASELINE config 5 (2^20-row inner STARK, outer circuit 2^19 rows): 16 proofs per step, steps only enqueued."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    import numpy as np, torch
    import __graft_entry__ as ge
    p25 = ge.load_package()
    if lib != "base":
        sys.modules["plonky25_amd.binding"].lib_path = lib
    p25.device_init(0)
    dev = torch.device("cuda", 0)
    inp, cfg = p25.p3_prove_fibonacci(20, 100, 16, threads=os.cpu_count())
    c = p25.Circuit.build_p3_verifier(cfg); c.digest()
    pw, B = int(c.info.proof_words), 16
    d_in = torch.from_numpy(np.stack([inp] * B).view(np.int64)).to(dev)
    d_seeds = torch.arange(B, dtype=torch.int64, device=dev)
    d_p = torch.zeros((B, pw), dtype=torch.int64, device=dev)
    d_s = torch.zeros((steps + 1, B), dtype=torch.int32, device=dev)
    for k in range(steps + 1):
        if k == 1:
            c.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        c.prove_dev(d_in.data_ptr(), B, d_seeds.data_ptr(), d_p.data_ptr(), pw, d_s[k].data_ptr())
    c.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"config5_proofs_per_s": round(B * steps / dt, 3), "ok": bool((d_s.cpu().numpy() == 0).all())}
    for i in range(2):
        _p, _s, tm = c.prove(inp, seeds=[i], timings=True)
    out["single"] = {k: round(v, 2) for k, v in tm.as_dict().items()}
    print("AB " + json.dumps(out), flush=True)


def child(lib, batch, steps, agg, pipe=0, tstreams="4,2,1"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    import numpy as np, torch
    import __graft_entry__ as ge
    p25 = ge.load_package()
    if lib != "base":
        sys.modules["plonky25_amd.binding"].lib_path = lib
    p25.device_init(0)
    dev = torch.device("cuda", 0)
    with open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")) as f:
        base, cfg = p25.p3_proof_from_json(f.read())
    variants = [base] + [p25.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)[0] for v in range(1, 8)]
    c = p25.Circuit.build_p3_verifier(cfg)
    if os.environ.get("P25_AB_STREAMS"):                 # proofs in flight (p25_circuit_set_streams), per variant: name=lib@P25_AB_STREAMS=20
        c.set_streams(int(os.environ["P25_AB_STREAMS"]))
    c.digest()
    pw = int(c.info.proof_words)
    d_in = torch.from_numpy(np.stack([variants[i % 8] for i in range(batch)]).view(np.int64)).to(dev)
    d_seeds = torch.arange(batch, dtype=torch.int64, device=dev)
    d_p = [torch.zeros((batch, pw), dtype=torch.int64, device=dev) for _ in range(2)]
    d_s = torch.zeros((steps + 1, batch), dtype=torch.int32, device=dev)
    out = {}
    for k in range(steps + 1):
        if k == 1:
            c.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        c.prove_dev(d_in.data_ptr(), batch, d_seeds.data_ptr(), d_p[k & 1].data_ptr(), pw, d_s[k].data_ptr())
    c.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["proofs_per_s"] = round(batch * steps / dt, 2)
    out["ok"] = bool((d_s.cpu().numpy() == 0).all())
    tm = None
    for i in range(4):
        _p, _s, tm = c.prove(variants[i], seeds=[i], timings=True)
    out["single"] = {k: round(v, 3) for k, v in tm.as_dict().items()}
    if agg:
        from plonky25_amd import aggregate as ag
        leaves = d_p[steps & 1][:agg].cpu().numpy().view(np.uint64)
        f = ag.fold(c, [leaves[i] for i in range(agg)], arity=8)
        out["agg"] = {"tree_s": round(f["tree_s"], 4), "levels_ms_per_proof": [l["ms_per_proof"] for l in f["levels"]]}
        a1 = f["owned"][0]                        # the level-1 aggregator alone: per-phase device times of one proof
        grp = np.concatenate([leaves[i] for i in range(8)])
        for i in range(3):
            _p, _s, tm1 = a1.prove(grp, seeds=[i], timings=True)
        out["agg"]["single"] = {k: round(v, 3) for k, v in tm1.as_dict().items()}
    if pipe:   # the pipelined device-resident tree (bench.py's `aggregation.pipelined`): warm-up steps + `pipe` timed steps
        from plonky25_amd import aggregate as ag
        ls = tuple(int(x) for x in tstreams.split(",")) if tstreams != "0" else None
        tree = ag.DeviceTree(c, batch, int(os.environ.get("P25_AB_ARITY", "13")), dev, leaf_batch=batch, level_streams=ls)
        nst = len(tree.levels) + 1 + pipe
        d_sp = torch.zeros((nst, batch), dtype=torch.int32, device=dev)

        def leaves(buf, j):
            c.prove_dev(d_in.data_ptr(), batch, d_seeds.data_ptr(), buf.data_ptr(), pw, d_sp[j].data_ptr())
        for _ in range(len(tree.levels) + 1):      # fill the pipeline: every level has run once
            tree.step(leaves)
        tree.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(pipe):                      # steady state: every host step = one leaf batch + one instance of every level
            tree.step(leaves)
        tree.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tree.flush(); tree.sync(); torch.cuda.synchronize()
        _root, rok = tree.root(tree.leaf_steps - 1)
        out["pipe"] = {"steps": pipe, "leaf_eq_proofs_per_s": round(batch * pipe / dt, 2), "ms_per_step": round(dt / pipe * 1e3, 1),
                       "ok": bool(rok and (d_sp.cpu().numpy() == 0).all())}
    print("AB " + json.dumps(out), flush=True)


def main():
    a = sys.argv[1:]
    if a and a[0] == "--child5":
        return child_config5(a[1], int(a[2]))
    if a and a[0] == "--child":
        return child(a[1], int(a[2]), int(a[3]), int(a[4]), int(a[5]), a[6])
    batch, steps, rounds, agg, pipe, libs, tstreams, cfg5 = 256, 3, 2, 0, 0, [], "0", False
    i = 0
    while i < len(a):
        if a[i] == "--batch": batch = int(a[i + 1]); i += 2
        elif a[i] == "--steps": steps = int(a[i + 1]); i += 2
        elif a[i] == "--rounds": rounds = int(a[i + 1]); i += 2
        elif a[i] == "--agg": agg = int(a[i + 1]); i += 2
        elif a[i] == "--pipe": pipe = int(a[i + 1]); i += 2
        elif a[i] == "--config5": cfg5 = True; i += 1
        elif a[i] == "--tree-streams": tstreams = a[i + 1]; i += 2
        elif a[i] == "--hwq": os.environ["GPU_MAX_HW_QUEUES"] = a[i + 1]; i += 2     # inherited by the children
        else:
            # name=path[@ENV=VAL[,ENV=VAL...]]: the environment of that variant's child processes (experiment knobs of a
            # tools/knobs_build.sh library, e.g. q2=tools/build/variants/libp25_knobs.so@P25_X_Q_LDS_PAD=40960)
            name, _, rest = a[i].partition("=")
            path, _, envs = rest.partition("@")
            libs.append((name, path or "base", tstreams, dict(kv.split("=", 1) for kv in envs.split(",") if kv))); i += 1
    for r in range(rounds):
        for name, path, ts, env in libs:
            argv = (["--child5", path, str(steps)] if cfg5 else ["--child", path, str(batch), str(steps), str(agg), str(pipe), ts])
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv,
                               capture_output=True, text=True, env=dict(os.environ, **env))
            line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
            print(f"round {r} {name:16s} hwq {os.environ.get('GPU_MAX_HW_QUEUES', '24'):3s} tree-streams {ts:8s} {line[0][3:] if line else 'FAILED ' + p.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
