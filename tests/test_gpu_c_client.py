"""GPU: a plain-C program (tests/c_abi/example.c) drives the whole path through include/p25.h --
header usable from C, symbols link, statuses and determinism as documented."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_client(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    libdir = os.path.join(ROOT, "plonky2.5_amd")
    exe = str(tmp_path / "c_client")
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "c_abi", "example.c"), "-L" + libdir, "-lp25",
                        "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "C CLIENT OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_c_client_chains_two_circuits_on_the_device(tmp_path):
    """tests/c_abi/chain.c: leaf proofs -> aggregation circuit through device buffers, ordered by p25_circuit_mark /
    p25_circuit_wait_mark, from plain C (HIP runtime for the buffers only): byte-equal to the host-buffer path."""
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    libdir = os.path.join(ROOT, "plonky2.5_amd")
    exe = str(tmp_path / "c_chain")
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__",
                        "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "chain.c"),
                        "-L" + libdir, "-lp25", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                        "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "C CHAIN OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_c_client_gathers_over_a_library_owned_rccl_communicator(tmp_path):
    """tests/c_abi/gather.c: p25_comm_unique_id / p25_comm_init / p25_gather_proofs / p25_comm_barrier / p25_comm_max_f64 from
    plain C -- the final aggregation step of north_star behind the C ABI (no torch in the process), a world of one rank on the
    one GPU: three pipelined steps gathered on the communicator's stream behind p25_circuit_mark, bytes == the host path."""
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    libdir = os.path.join(ROOT, "plonky2.5_amd")
    exe = str(tmp_path / "c_gather")
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__",
                        "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "gather.c"),
                        "-L" + libdir, "-lp25", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                        "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)     # the client checks that the library's own request is the one in force
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "C GATHER OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_c_client_multi_rank_gather_through_a_test_double_of_librccl(tmp_path):
    """tests/c_abi/gather_multi.c: the MULTI-RANK half of p25_gather_proofs on the one GPU there is -- every rank a thread, librccl
    replaced by the test double tests/c_abi/fake_rccl.cpp (RCCL refuses two ranks on one device): who sends what to whom, receive
    offsets of uneven and empty shards, roots other than rank 0, gathers in flight on one communicator, the max reduction and the
    barrier over 2-5 ranks.  Every gathered word is checked; the double's counters prove it carried the traffic."""
    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    if not gcc or not gxx:
        pytest.skip("gcc / g++ not available")
    libdir = os.path.join(ROOT, "plonky2.5_amd")
    fake_dir = tmp_path / "fake"
    fake_dir.mkdir()
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-shared", "-fPIC",
                        "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "fake_rccl.cpp"),
                        "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,librccl.so.1",
                        "-o", str(fake_dir / "librccl.so.1")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = str(tmp_path / "c_gather_multi")
    r = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__", "-D_GNU_SOURCE",
                        "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "gather_multi.c"),
                        "-L" + libdir, "-lp25", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-ldl", "-Wl,-rpath," + libdir,
                        "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_LIBRARY_PATH=str(fake_dir) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "C GATHER MULTI OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
