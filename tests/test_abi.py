"""CPU: libp25.so loads and exports every symbol include/p25.h declares (no compute without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "p25.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(p25_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(p25):
    lib = p25.lib()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/p25.h but not exported"
        assert n in p25.EXPORTED_SYMBOLS, f"{n} missing from the Python binding table"
    assert b"gfx950" in lib.p25_version()


def test_no_device_fails_loudly(p25):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    with pytest.raises(p25.P25Error) as e:
        p25.poseidon_permute(np.zeros(12, dtype=np.uint64))
    assert e.value.status == 2  # P25_ERR_NO_DEVICE: no CPU fallback
