#!/usr/bin/env python3
"""What is resident when: from a rocprofv3 --kernel-trace CSV of a batch run (16 proofs in flight), the share of wall
time with k kernels of each class active, and each class's summed in-pipeline residency against its launch count.

Classes: hash (k_hash_leaves*, k_tree_level -- the bulk VALU work), ntt (k_ntt_tile), quot (k_quotient*), chain (single-wave
or few-wave latency chains: transcript, witness levels, cooperative tree tops, PoW, FRI tails), other.

usage: timeline.py kernel_trace.csv [t0_frac t1_frac]    (the window of the trace analysed, default 0.25 0.9: steady state)"""
import csv
import sys
from collections import Counter, defaultdict


def klass(name):
    if "k_hash_leaves" in name or "k_tree_level(" in name or name.endswith("k_tree_level") or "k_tree_level " in name:
        return "hash"
    if "k_tree_level_coop" in name or "k_tree_top_coop" in name or "k_transcript" in name or "k_witgen" in name \
            or "k_pow" in name or "k_fri_final" in name or "k_queries" in name:
        return "chain"
    if "k_tree_level" in name:
        return "hash"
    if "k_ntt_tile" in name:
        return "ntt"
    if "k_quotient" in name:
        return "quot"
    return "other"


def main():
    path = sys.argv[1]
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[2][0].isdigit() else (0.25, 0.9)
    rd = csv.reader(open(path))
    hdr = next(rd)
    ki, si, ei = hdr.index("Kernel_Name"), hdr.index("Start_Timestamp"), hdr.index("End_Timestamp")
    qi = hdr.index("Queue_Id") if "Queue_Id" in hdr else None
    sti = hdr.index("Stream_Id") if "Stream_Id" in hdr else None
    raw = list(rd)
    rows = [(int(r[si]), int(r[ei]), klass(r[ki]), r[ki]) for r in raw]
    if "--dump" in sys.argv:      # compact copy for offline analysis: start, end, class, queue, stream, short kernel name
        import gzip
        out = sys.argv[sys.argv.index("--dump") + 1]
        t00 = min(r[0] for r in rows)
        with gzip.open(out, "wt") as g:
            for r, (s, e, k, name) in zip(raw, rows):
                short = name.split("(")[0].replace("p25::", "").replace("void ", "")[:32]
                g.write(f"{s - t00},{e - t00},{k},{r[qi] if qi is not None else -1},{r[sti] if sti is not None else -1},{short}\n")
    t_min, t_max = min(r[0] for r in rows), max(r[1] for r in rows)
    w0, w1 = t_min + f0 * (t_max - t_min), t_min + f1 * (t_max - t_min)
    ev = []
    resid = defaultdict(float)
    count = Counter()
    per_kernel = defaultdict(lambda: [0, 0.0])
    for s, e, k, name in rows:
        s2, e2 = max(s, w0), min(e, w1)
        if e2 <= s2:
            continue
        ev.append((s2, 1, k))
        ev.append((e2, -1, k))
        resid[k] += e2 - s2
        count[k] += 1
        short = name.split("(")[0].replace("p25::", "").replace("void ", "")[:40]
        per_kernel[short][0] += 1
        per_kernel[short][1] += e2 - s2
    ev.sort()
    active = Counter()
    hist = defaultdict(lambda: defaultdict(float))   # class -> number active -> time
    combos = defaultdict(float)
    total_hist = defaultdict(float)
    last = w0
    for t, d, k in ev:
        dt = t - last
        if dt > 0:
            for c in ("hash", "ntt", "quot", "chain", "other"):
                hist[c][min(active[c], 8)] += dt
            key = ("hash>=3" if active["hash"] >= 3 else f"hash={active['hash']}",
                   "mem>=3" if active["ntt"] + active["quot"] >= 3 else f"mem={active['ntt'] + active['quot']}")
            combos[key] += dt
            total_hist[min(sum(active.values()), 24)] += dt
        active[k] += d
        last = t
    W = w1 - w0
    print(f"window {W / 1e6:.1f} ms of a {(t_max - t_min) / 1e6:.1f} ms trace; kernels in window: {sum(count.values())}")
    print("share of wall time with k kernels of the class ACTIVE (dispatched, not finished); 8 = 8 or more")
    print("class   " + "".join(f"{k:>7d}" for k in range(9)) + "   mean active   launches   residency ms (sum)")
    for c in ("hash", "ntt", "quot", "chain", "other"):
        tot = sum(hist[c].values()) or 1.0
        mean = sum(k * v for k, v in hist[c].items()) / tot
        print(f"{c:7s} " + "".join(f"{100 * hist[c].get(k, 0.0) / tot:7.1f}" for k in range(9)) +
              f"   {mean:11.2f}   {count[c]:8d}   {resid[c] / 1e6:10.1f}")
    print("kernels of ANY class active at once, % of wall time: " +
          " ".join(f"{k}:{100 * v / W:.1f}" for k, v in sorted(total_hist.items())))
    print("joint: bulk hash kernels active x memory-waiting kernels (ntt + quotient) active, % of wall time")
    for key in sorted(combos):
        print(f"  {key[0]:8s} {key[1]:7s} {100 * combos[key] / W:6.2f}")
    print("per kernel in the window: launches, mean in-pipeline duration ms")
    for name, (n, tot) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {name:42s} {n:6d} {tot / n / 1e6:9.3f}")


if __name__ == "__main__":
    main()
