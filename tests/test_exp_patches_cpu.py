"""CPU: the experiment patches under tools/exp/ still apply to the product sources (they are how any measured-and-off switch of
rounds 1-5 is brought back for an A/B run: tools/exp/apply.sh; the product itself carries none), and the product sources really
are free of experiment switches: the only preprocessor conditionals are the host / device split."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CURRENT = ["switches_kernels_hash.patch", "switches_kernels_ntt.patch", "switches_kernels_quotient.patch",
           "switches_arith_sched.patch"]


def test_switch_patches_apply_to_the_current_sources():
    if not shutil.which("patch"):
        pytest.skip("patch not available")
    apply_sh = os.path.join(ROOT, "tools", "exp", "apply.sh")
    r = subprocess.run([apply_sh, "pytest_all"] + ["tools/exp/" + p for p in CURRENT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FAILED" not in r.stdout + r.stderr, (r.stdout[-1500:], r.stderr[-1500:])
    scratch = r.stdout.strip().splitlines()[-1]
    text = "".join(open(os.path.join(scratch, f)).read() for f in ("kernels_hash.hip", "kernels_ntt.hip", "kernels_quotient.hip",
                                                                      "gl.h", "gl_lazy.h", "poseidon.h", "prover.hip", "kernels.h"))
    for switch in ("P25_LEAF_MX", "P25_TREE_MINW", "P25_HASH_PERSIST", "P25_NTT_PERSIST", "P25_Q_WAVES", "P25_Q_MERGE_PERM",
                   "P25_PROFILE_GATE_MASK", "P25_ASM_MUL", "P25_PARTIAL3", "P25_STREAM_POOL", "P25_EXPERIMENT_KNOBS"):
        assert switch in text, switch
    r = subprocess.run([apply_sh, "pytest_lazy", "tools/exp/lazy_contract_check.patch"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FAILED" not in r.stdout + r.stderr, (r.stdout[-1500:], r.stderr[-1500:])
    shutil.rmtree(os.path.join(ROOT, "tools", "build", "exp", "pytest_all"), ignore_errors=True)
    shutil.rmtree(os.path.join(ROOT, "tools", "build", "exp", "pytest_lazy"), ignore_errors=True)


def test_product_sources_carry_no_experiment_switches():
    csrc = os.path.join(ROOT, "plonky2.5_amd", "csrc")
    allowed = re.compile(r"^\s*#\s*(if|ifdef|ifndef|elif)\b(.*)$")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h", ".cpp")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, fn)), 1):
            m = allowed.match(line)
            if not m:
                continue
            cond = m.group(2).strip()
            ok = cond in ("defined(__HIPCC__)", "defined(__HIP_DEVICE_COMPILE__)", "__HIPCC__") or \
                (m.group(1) == "ifdef" and cond == "__HIPCC__")
            assert ok, f"{fn}:{ln}: conditional on something other than the host / device split: {line.strip()}"
        src = open(os.path.join(csrc, fn)).read()
        assert "getenv" not in src or fn == "capi.hip", f"{fn} reads the environment"     # capi.hip: GPU_MAX_HW_QUEUES only
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "-D" not in mk, "the product builds with zero -D flags"
    assert '#include "poseidon_mfma.h"' not in open(os.path.join(csrc, "kernels_hash.hip")).read()
