"""Like fd_sampler.py, but logs every change to a file (kept if the run is killed) and looks at fds, maps and KFD's own process list."""
import os, subprocess, sys, time
log = open(sys.argv[1], "w")
cmd = sys.argv[2]
def scan():
    found = {}
    for pid in os.listdir("/proc"):
        if not pid.isdigit(): continue
        hits = set()
        try:
            for fd in os.listdir(f"/proc/{pid}/fd"):
                try:
                    t = os.readlink(f"/proc/{pid}/fd/{fd}")
                except OSError:
                    continue
                if "kfd" in t or "/dev/dri" in t:
                    hits.add("fd:" + t)
            with open(f"/proc/{pid}/maps") as f:
                for line in f:
                    if "kfd" in line or "renderD" in line:
                        hits.add("map:" + line.split()[-1])
        except OSError:
            continue
        if hits:
            try:
                cl = open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:70]
            except OSError:
                cl = "?"
            found[pid] = (cl, tuple(sorted(hits)))
    try:
        kfdp = sorted(os.listdir("/sys/class/kfd/kfd/proc"))
    except OSError as e:
        kfdp = [str(e)[:40]]
    return found, kfdp
p = subprocess.Popen(cmd, shell=True)
t0, last = time.time(), None
while p.poll() is None and time.time() - t0 < 300:
    f, kfdp = scan()
    state = (tuple(sorted((pid, v) for pid, v in f.items())), tuple(kfdp))
    if state != last:
        last = state
        log.write(f"t={time.time() - t0:.2f} gpu-open processes: {len(f)} kfd-proc: {kfdp}\n")
        for pid, (cl, hits) in sorted(f.items()):
            log.write(f"   {pid} {cl} {hits}\n")
        log.flush()
    time.sleep(0.02)
log.write(f"done rc={p.returncode} t={time.time() - t0:.1f}\n")
