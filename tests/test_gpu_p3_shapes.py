"""GPU parity on verifier circuits for other plonky3 proof shapes (inputs from the native p3 prover):
different circuit sizes (2^11, 2^12, 2^17 rows), FRI schedules and final-polynomial lengths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,queries,pow_bits", [(3, 4, 8), (4, 10, 8), (8, 100, 16)])
def test_gpu_equals_oracle_on_shape(gpu, oracle, log_n, queries, pow_bits):
    inp, cfg = gpu.p3_prove_fibonacci(log_n, queries, pow_bits)
    c = gpu.Circuit.build_p3_verifier(cfg)
    oc = oracle.load_circuit(c.to_blob())
    dg, capg = c.digest()
    do, capo = oc.digest()
    assert (dg == do).all() and (capg == capo).all()
    other, _ = gpu.p3_prove_fibonacci(log_n, queries, pow_bits, pow_start=int(inp[-1 - 0]) if False else 1 << 20)
    proofs, st = c.prove(np.stack([inp, other]), seeds=[3, 4])
    assert st.tolist() == [0, 0]
    po, sto, _tm, msg = oc.prove(inp, seed=3)
    assert sto == 0, msg
    diff = np.nonzero(proofs[0] != po)[0]
    assert diff.size == 0, diff[:8]
    assert oc.verify(proofs[1], dg, capg)[0] == 0


def test_two_distinct_fib64_proofs_in_one_batch(gpu, fib_circuit, fib_oracle, fib_inputs):
    inp2, _ = gpu.p3_prove_fibonacci(6, 100, 16, pow_start=103885)
    proofs, st = fib_circuit.prove(np.stack([fib_inputs, inp2]), seeds=[1, 1])
    assert st.tolist() == [0, 0]
    assert (proofs[0] != proofs[1]).any()
    dg, capg = fib_circuit.digest()
    for p in proofs:
        assert fib_oracle.verify(p, dg, capg)[0] == 0
