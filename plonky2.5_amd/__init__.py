"""plonky2.5_amd -- host-side Python binding of libp25 (MI355X-native plonky2 prover for the
plonky3-verifier circuit).  Pure ctypes over the C ABI in include/p25.h; numpy for host buffers,
torch only as plumbing for device memory / streams in bench.py.

The directory name contains a dot, so import it through `tests/conftest.py::load_p25()` /
`__graft_entry__.load_package()` (importlib with an explicit module name `plonky25_amd`).

There is NO CPU fallback: every compute call goes to the HIP library and raises `P25Error`
if libp25.so is missing or no HIP device is present.
"""
from .binding import *  # noqa: F401,F403
from .binding import __all__  # noqa: F401
from . import dist  # noqa: E402,F401
from . import aggregate  # noqa: E402,F401
