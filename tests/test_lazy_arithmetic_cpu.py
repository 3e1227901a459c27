"""CPU: the plain C++ definitions of the lazy (non-canonical) Goldilocks operations in gl_lazy.h / ntt16.h -- what the
hand-written gfx950 sequences are compared with on the device (tools/asmcheck.hip, tests/test_gpu_asmcheck.py) -- against
128-bit integer arithmetic mod p: all pairs of boundary values (where the double wraps occur), random operands biased to the
top of the range, every shift exponent, the wrapped decrements inside vanishing products, the lazy 16-point network against
the naive DFT (tests/native/lazy_defs.cpp)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_lazy_definitions_agree_with_wide_integer_arithmetic(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "lazy_defs")
    r = subprocess.run([gxx, "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "plonky2.5_amd", "csrc"),
                        os.path.join(ROOT, "tests", "native", "lazy_defs.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LAZY DEFS OK" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
