"""GPU: a short, seeded run of tools/fuzz_parity.py -- random AIR families (degree 1..5, FriConfig.log_blowup 1..3), trace heights, query counts, PoW bits,
gadget operands and corrupted inputs; every status and every proof byte must equal the oracle's."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_seeded_fuzz_run_agrees_with_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "40", "7"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("FUZZ OK") and " 0 differences" in last, last
    assert int(last.split()[2]) >= 50, last          # a run that proved next to nothing is not a pass
