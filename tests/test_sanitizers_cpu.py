"""CPU: AddressSanitizer + UBSan over the product's host code and the oracle (tests/sanitize/).
GPU sanitizers are not available on the pool, so the CPU build is where memory errors are hunted."""
import glob
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_code_and_oracle_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    csrc = os.path.join(ROOT, "plonky2.5_amd", "csrc")
    ora = os.path.join(ROOT, "oracle")
    srcs = [os.path.join(ROOT, "tests", "sanitize", "host_and_oracle.cpp")]
    srcs += [os.path.join(csrc, f) for f in ("builder.cpp", "p3_circuit.cpp", "p3_prover.cpp", "circuit_io.cpp",
                                             "circuit_bytes.cpp", "witness_program.cpp", "recursion.cpp")]
    srcs += sorted(glob.glob(os.path.join(ora, "*.cpp")))
    exe = str(tmp_path / "san")
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer", "-I" + csrc, "-I" + ora, "-I" + os.path.join(ROOT, "include")]
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        extra = ["-mavx512f", "-mavx512dq"] if os.path.basename(src) == "ref_quotient_x8.cpp" else []   # as oracle/Makefile
        return obj, subprocess.run([gxx, *flags, *extra, "-c", src, "-o", obj], capture_output=True, text=True, timeout=900)

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        results = list(ex.map(compile_one, srcs))
    for obj, r in results:
        assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([gxx, "-fsanitize=address,undefined", *[o for o, _ in results], "-lpthread", "-o", exe],
                       capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0 and "SANITIZE OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
