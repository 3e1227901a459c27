"""GPU: the hand-written gfx950 sequences (field multiply / multiply-add, the MDS layer with folded constants, the fused
permutation, and the lazy-arithmetic operations of gl_lazy.h: butterfly pairs, single-correction adds / subs, products by
powers of two, the 16-point NTT network) against their plain C++ forms on the device, over edge cases and 10^6 random operands
(tools/asmcheck.hip, built by __graft_entry__.build())."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_hand_written_sequences_match_their_cpp_forms():
    exe = os.path.join(ROOT, "tools", "build", "asmcheck")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASMCHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    assert "mul_nc_asm: 0 mismatches" in r.stdout and "mad_nc_asm: 0 mismatches" in r.stdout
    # the lazy-arithmetic sequences of gl_lazy.h (round 5): sum / difference pairs, single operations, shifts, the 16-point network
    for line in ("bfly_nc_asm: 0 mismatches", "vanishing products: 0 mismatches", "shl_nc_asm (e = 0..95): 0 mismatches",
                 "vs naive DFT: 0 mismatches"):
        assert line in r.stdout, r.stdout[-2000:]
