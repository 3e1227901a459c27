"""GPU: the wave-wide Poseidon permutation with its full-round MDS layers on the matrix cores (poseidon_mfma.h; a build
option of the leaf / tree kernels, off by default) against the VALU permutation, on the device: one MDS layer against
the host definition, the raw accumulators against the assumed operand / result layouts of v_mfma_i32_32x32x32_i8, and
three chained permutations over 2^19 states in each `rows` mode (tools/mdsbench.hip, built by __graft_entry__.build())."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_mfma_permutation_is_bit_identical_to_the_valu_permutation():
    exe = os.path.join(ROOT, "tools", "build", "mdsbench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    out = r.stdout
    assert r.returncode == 0, out[-2000:] + r.stderr[-500:]
    assert "pair layout after the swaps: 0 mismatches" in out
    assert "raw accumulators (u = 0, batch A): 0 mismatches" in out
    assert len(re.findall(r"mfma vs valu kernel: 0 words differ of \d+; mfma vs host definition \(4096 states\): 0 differ", out)) == 2, out[-2000:]
    assert "one layer (round 0): 0 words differ" in out and "rounds 0-2 + sbox 3: 0 words differ" in out
    for rows in ("fff", "00f", "f00"):
        assert f"permutation x3, rows {rows}: permute_wave vs permute_dev: 0 words differ" in out, out[-2000:]
    assert "differ" in out and not re.search(r": [1-9]\d* (words|dwords) differ", out), out[-2000:]
