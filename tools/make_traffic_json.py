#!/usr/bin/env python3
"""profiles/pmc_hash_leaves.json from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh.
usage: make_traffic_json.py <tag>   (reads gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.json)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
f = json.load(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_FETCH_SIZE.json")))
w = json.load(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_WRITE_SIZE.json")))
K = "p25::k_hash_leaves_wide"
fetch_kb, write_kb = f[K]["FETCH_SIZE"] / f[K]["calls"], w[K]["WRITE_SIZE"] / w[K]["calls"]
n_big, width = 1 << 19, 135
import hashlib, subprocess
def kernel_source_sha():
    """sha256 over the sources of the dominant kernel: bench.py reports the PMC traffic only while these are unchanged."""
    h = hashlib.sha256()
    for f in ("kernels_hash.hip", "poseidon.h", "poseidon_p3r.h", "gl.h"):
        h.update(open(os.path.join(ROOT, "plonky2.5_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
out = {
    "kernel_source_sha": kernel_source_sha(),
    "head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) "
              "-- python3 tools/prove_one.py 4 (per-proof figures)   [tools/collect_profiles.sh " + tag + "]",
    "kernel": "k_hash_leaves_wide on the 2^19 x 135 wires LDE (one launch per proof)",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B: x2 (MI355X_MICROARCH.md, HBM section); "
                  "calibrated on this very kernel: it reads the 566,231,040-B LDE exactly once and the counter "
                  "shows half of that; WRITE_SIZE is exact (16,777,216 B of digests)",
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "algorithmic_bytes_per_launch": n_big * width * 8 + n_big * 32,
    "per_kernel_hbm_MB_per_proof": {k.replace("p25::", ""): round((f[k].get("FETCH_SIZE", 0) * 2 + w.get(k, {}).get("WRITE_SIZE", 0)) * 1024 / 1e6, 1)
                                    for k in f if f[k].get("FETCH_SIZE", 0) * 2 + w.get(k, {}).get("WRITE_SIZE", 0) > 1000},
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_hash_leaves.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
