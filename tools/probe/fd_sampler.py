"""Samples every process of the box for open GPU device nodes while N children run a command; prints what was seen.
usage: fd_sampler.py N 'python -c "import torch"' """
import os, subprocess, sys, time, collections
n, cmd = int(sys.argv[1]), sys.argv[2]
seen = collections.Counter(); peak = 0; peak_detail = None
def scan():
    found = {}
    for pid in os.listdir("/proc"):
        if not pid.isdigit(): continue
        try:
            fds = os.listdir(f"/proc/{pid}/fd")
        except OSError:
            continue
        hits = []
        for fd in fds:
            try:
                t = os.readlink(f"/proc/{pid}/fd/{fd}")
            except OSError:
                continue
            if "kfd" in t or "/dev/dri" in t:
                hits.append(t)
        if hits:
            try:
                cl = open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:80]
            except OSError:
                cl = "?"
            found[pid] = (cl, sorted(set(hits)))
    return found
procs = [subprocess.Popen(cmd, shell=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(n)]
t0 = time.time()
while any(p.poll() is None for p in procs) and time.time() - t0 < 200:
    f = scan()
    for pid, (cl, hits) in f.items():
        seen[(cl, tuple(hits))] += 1
    if len(f) > peak:
        peak, peak_detail = len(f), f
    time.sleep(0.02)
print("elapsed", round(time.time() - t0, 1), "peak processes with a GPU node open:", peak)
for (cl, hits), c in seen.most_common(12):
    print(c, cl, hits)
