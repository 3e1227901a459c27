import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
ARTIFACT = os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")


def free_port():
    """A rendezvous port the OS hands out (as bench.py does): a fixed number can collide with another job on the box."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_p25():
    """Import the package directory `plonky2.5_amd/` (dot in the name) as module `plonky25_amd`."""
    name = "plonky25_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "plonky2.5_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_binding import Oracle, OracleCircuit, splitmix_field  # noqa: E402,F401  (the checker)


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def p25():
    return load_p25()


@pytest.fixture(scope="session")
def gpu(p25):
    p25.device_init(0)
    return p25


@pytest.fixture(scope="session")
def fib_circuit(p25):
    """The plonky3-verifier circuit for the fib-64 artifact, built by the product's host code."""
    return p25.Circuit.build_p3_verifier(p25.P3Config.fib64())


@pytest.fixture(scope="session")
def fib_blob(fib_circuit):
    return fib_circuit.to_blob()


@pytest.fixture(scope="session")
def fib_oracle(oracle, fib_blob):
    return oracle.load_circuit(fib_blob)


@pytest.fixture(scope="session")
def fib_inputs():
    import p3json
    inp, _shape = p3json.load(ARTIFACT)
    return inp
