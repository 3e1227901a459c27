#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV (or the grep'd rows of one kernel + the header line): the dominant
kernel's average duration over ALL its launches, over the launches of bench.py's timed region, and over the
last 8 (bench.py's single-proof passes, where the kernel has the GPU to itself = roofline.avg_launch_ms).
usage: rocprof_kernel_split.py <kernel_trace.csv | rows.csv header.csv> [kernel substring]"""
import csv, sys
args = [a for a in sys.argv[1:]]
name = "k_hash_leaves_wide"
if len(args) >= 2 and args[1].endswith(".csv"):
    hdr = next(csv.reader(open(args[1])))
    rows = list(csv.reader(open(args[0])))
    if len(args) > 2: name = args[2]
else:
    rd = list(csv.reader(open(args[0])))
    hdr, rows = rd[0], rd[1:]
    if len(args) > 1: name = args[1]
ki, si, ei = hdr.index("Kernel_Name"), hdr.index("Start_Timestamp"), hdr.index("End_Timestamp")
sel = sorted(((int(r[si]), int(r[ei])) for r in rows if name in r[ki]))
d = [(e - s) / 1e6 for s, e in sel]
print(f"kernel {name}: {len(d)} launches, average {sum(d)/len(d):.3f} ms")
if len(d) > 8:
    print(f"  last 8 launches (one proof in flight, GPU to itself): average {sum(d[-8:])/8:.3f} ms  min {min(d[-8:]):.3f} max {max(d[-8:]):.3f}")
    rest = d[:-8]
    print(f"  the other {len(rest)} launches (warm-up + timed region, up to 16 proofs in flight): average {sum(rest)/len(rest):.3f} ms")
