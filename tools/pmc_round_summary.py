#!/usr/bin/env python3
"""Per-kernel tables (VALU instructions, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE against the algorithmic bytes, wait share)
from the PMC json files of tools/collect_profiles.sh / tools/pmc_config5.sh:
    python3 tools/pmc_round_summary.py r04_z        -> profiles/<tag>_pmc_summary.txt, profiles/<tag>_config5_pmc_summary.txt
(reads profiles/<tag>_pmc_*.json and profiles/<tag>_config5_pmc_*.json)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def load(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else {}


def pick(d, name, counter):
    """Sum over the kernels whose short name starts with `name` (k_hash_leaves does not swallow k_hash_leaves_wide)."""
    tot = 0.0
    for n, x in d.items():
        if n == "_meta":
            continue
        short = n.replace("p25::", "").replace("void ", "")
        if name == "k_hash_leaves" and "wide" in short:
            continue
        if name == "k_quotient" and "rec" in short:
            continue
        if short.startswith(name):
            tot += x.get(counter, 0.0)
    return tot


def total(d, counter):
    return sum(x.get(counter, 0.0) for n, x in d.items() if n != "_meta")


def table(prefix, rows, unit, scale, header, footer, out_name, alg_total=None):
    v, f, w, a = (load(f"{prefix}_pmc_{c}.json") for c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES"))
    out = list(header)
    out.append(f"{'kernel':26s} {'VALU (' + unit[0] + ' wave-instr)':>20s} {'HBM ' + unit[1] + ' = 2 x FETCH + WRITE':>34s} {'algorithmic':>12s} {'ratio':>6s} {'SQ_WAIT_ANY / SQ_WAVE_CYCLES':>30s}")
    for label, key, alg in rows:
        vi = pick(v, key, "SQ_INSTS_VALU") / scale[0]
        fe = pick(f, key, "FETCH_SIZE") * 1024 * 2 / scale[1]
        wr = pick(w, key, "WRITE_SIZE") * 1024 / scale[1]
        wa, wc = pick(a, key, "SQ_WAIT_ANY"), pick(a, key, "SQ_WAVE_CYCLES")
        out.append(f"{label:26s} {vi:20.2f} {fe:16.1f} + {wr:6.1f} = {fe + wr:7.1f} {('%.1f' % alg) if alg else '':>12s} "
                   f"{('%.1f' % ((fe + wr) / alg)) if alg else '':>6s} {('%.0f %%' % (100 * wa / wc)) if wc else '':>30s}")
    tf, tw = total(f, "FETCH_SIZE") * 1024 * 2 / scale[1], total(w, "WRITE_SIZE") * 1024 / scale[1]
    out.append(f"{'whole proof':26s} {total(v, 'SQ_INSTS_VALU') / scale[0]:20.2f} {tf:16.1f} + {tw:6.1f} = {tf + tw:7.1f} "
               f"{('%.1f' % alg_total) if alg_total else '':>12s} {('%.1f' % ((tf + tw) / alg_total)) if alg_total else '':>6s}")
    meta = v.get("_meta", {})
    out.append(f"\n(kernel sources {meta.get('csrc_sha', '?')}, collected at {meta.get('head', '?')})")
    out += footer
    open(os.path.join(P, out_name), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


table(tag,
      [("k_hash_leaves_wide", "k_hash_leaves_wide", 583.0), ("k_ntt_tile (fwd + inv)", "k_ntt_tile", 905.0), ("k_quotient", "k_quotient", 1023.0),
       ("k_hash_leaves (2)", "k_hash_leaves", None), ("k_tree_level*", "k_tree_level", None), ("k_witgen_fill_wires", "k_witgen_fill_wires", 71.0)],
      ("M", "MB"), (1e6, 1e6),
      ["BASELINE config 3 kernels (fib-64 circuit: 2^16 rows x 135 wires, LDE 2^19), per proof, four proofs in flight:",
       "rocprofv3 --pmc {SQ_INSTS_VALU ... | FETCH_SIZE | WRITE_SIZE | SQ_WAVE_CYCLES SQ_WAIT_ANY ...} --kernel-trace -- python3 tools/prove_one.py 4",
       f"(tools/collect_profiles.sh {tag} pmc; separate passes; FETCH_SIZE x2 per the gfx950 correction)\n"],
      ["Round 3 (profiles/r03_zz_*): k_quotient 4,291 MB (4.2x), SQ_WAIT_ANY 49 % of 2,790 M wave cycles, 112 spilled VGPRs; VALU 3,970 M per proof.",
       "Round 4: no spills (column reads through buffer descriptors, profiles/r04_ab_quotient_variants.txt).  The VERDICT r3 targets",
       "`k_quotient <= 3x, SQ_WAIT_ANY < 35 %` are NOT met: the evaluators still re-read wire columns (~620 column reads per point for",
       "135 + 85 + 22 distinct), and what its waves wait for is the first loads of each evaluator, not spills any more."],
      f"{tag}_pmc_summary.txt", alg_total=3220.0)
print()
table(f"{tag}_config5",
      [("k_hash_leaves_wide", "k_hash_leaves_wide", 4.66), ("k_ntt_tile (fwd + inv)", "k_ntt_tile", 7.2), ("k_quotient", "k_quotient", 8.2),
       ("k_hash_leaves (2)", "k_hash_leaves", None), ("k_tree_level*", "k_tree_level", None)],
      ("G", "GB"), (1e9, 1e9),
      ["BASELINE config 5 (inner Fibonacci trace 2^20 rows -> outer circuit 2^19 rows x 135 wires, LDE 2^22), per proof, two proofs in flight:",
       "rocprofv3 --pmc {SQ_INSTS_VALU | FETCH_SIZE | WRITE_SIZE} --kernel-trace -- python3 tools/prove_one.py 2 --log-n 20",
       f"(tools/pmc_config5.sh {tag}; separate passes; FETCH_SIZE x2 per the gfx950 correction)\n"],
      ["Round 3 (profiles/r03_z_config5_pmc_summary.txt): k_ntt_tile 57.3 GB (8x), k_quotient 44.7 GB (5.5x), whole proof ~116 GB.",
       "Round 4: 16-wide tiles on the 512-point strided pass, factored pre-scale tables, 8-wide / 512-thread tiles on the 1024-point contiguous pass;",
       "what the NTT still moves over its two-pass floor (~2.6x: the LDE is written, re-read and re-written once) is pass 1 re-reading a tile's",
       "coefficients once per coset."],
      f"{tag}_config5_pmc_summary.txt", alg_total=25.7)
