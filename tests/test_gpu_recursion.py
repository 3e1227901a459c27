"""GPU: recursion -- circuits that verify this library's proofs, proved on the MI355X and checked
against the oracle: gate-level evaluator circuits, the recursive verifier of a small circuit, of a fib-64
plonky3-verifier proof (the batch item of BASELINE.json), and a 2-to-1 aggregation of two such proofs."""
import numpy as np
import pytest

from conftest import P, splitmix_field

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [3, 8, 9, 10, 11])
def test_gate_eval_circuits_gpu_equals_oracle(gpu, oracle, kind):
    wires = splitmix_field(270, seed=100 + kind).reshape(135, 2)
    consts = splitmix_field(4, seed=200 + kind).reshape(2, 2)
    pih = splitmix_field(4, seed=300 + kind)
    expect = oracle.eval_gate(kind, wires, consts, pih)
    inp = np.concatenate([wires.ravel(), consts.ravel(), pih, expect.ravel()])
    c = gpu.Circuit.build_gate_eval(kind)
    oc = oracle.load_circuit(c.to_blob())
    wrong = inp.copy()
    wrong[-1] = (int(wrong[-1]) + 1) % P
    proofs, st = c.prove(np.stack([inp, wrong]), seeds=[4, 4])
    assert st.tolist() == [0, 4]
    po, sto, _t, msg = oc.prove(inp, seed=4)
    assert sto == 0, msg
    assert (proofs[0] == po).all()
    dg, cap = c.digest()
    assert oc.verify(proofs[0], dg, cap)[0] == 0


def test_recursive_verifier_small_gpu_equals_oracle(gpu, oracle):
    inner = gpu.Circuit.build_gadget(0, 0)
    x, y = 0x0123456789ABCDEF % P, 0x0FEDCBA987654321 % P
    inp = np.array([x, y, (x & y) % P], dtype=np.uint64)
    inner_proofs, st = inner.prove(np.stack([inp, inp]), seeds=[5, 6])
    assert st.tolist() == [0, 0]
    outer = inner.build_recursive_verifier(1)               # verifier data from the GPU
    oo = oracle.load_circuit(outer.to_blob())
    bad = inner_proofs[1].copy()
    bad[70] = (int(bad[70]) + 1) % P
    outer_proofs, st = outer.prove(np.stack([inner_proofs[0], inner_proofs[1], bad]), seeds=[1, 2, 3])
    assert st.tolist() == [0, 0, 4]
    po, sto, _t, msg = oo.prove(inner_proofs[0], seed=1)
    assert sto == 0, msg
    diff = np.nonzero(outer_proofs[0] != po)[0]
    assert diff.size == 0, diff[:8]
    dg, cap = outer.digest()
    do, capo = oo.digest()
    assert (dg == do).all() and (cap == capo).all()
    assert oo.verify(outer_proofs[1], dg, cap)[0] == 0


@pytest.fixture(scope="module")
def fib_inner_proofs(gpu, fib_circuit, fib_inputs):
    alt, _ = gpu.p3_prove_fibonacci(6, 100, 16, pow_start=1 << 24)
    proofs, st = fib_circuit.prove(np.stack([fib_inputs, alt]), seeds=[11, 12])
    assert st.tolist() == [0, 0]
    return proofs


def test_recursive_verifier_of_fib64_proof(gpu, oracle, fib_circuit, fib_inner_proofs):
    """One level of recursion over the bench's batch item: the outer proof attests a fib-64 plonky3-verifier proof."""
    outer = fib_circuit.build_recursive_verifier(1)
    info = outer.info
    assert int(info.num_inputs) == int(fib_circuit.info.proof_words) == 19861
    print("recursive verifier of one fib-64 proof: 2^%d rows, %d rows used, %d generators, %d witness levels"
          % (int(info.degree_bits), int(info.num_rows_used), int(info.num_generators), int(info.witness_levels)))
    bad = fib_inner_proofs[1].copy()
    bad[5000] = (int(bad[5000]) + 1) % P
    proofs, st, tm = outer.prove(fib_inner_proofs[0], seeds=[1], timings=True)
    assert st.tolist() == [0]
    print("outer proof phase ms:", {k: round(v, 2) for k, v in tm.as_dict().items()})
    proofs3, st3 = outer.prove(np.stack([fib_inner_proofs[0], fib_inner_proofs[1], bad]), seeds=[1, 2, 3])
    assert st3.tolist() == [0, 0, 4]
    assert (proofs3[0] == proofs[0]).all()
    oo = oracle.load_circuit(outer.to_blob())
    dg, cap = outer.digest()
    do, capo = oo.digest()
    assert (dg == do).all() and (cap == capo).all()
    for k in (0, 1):
        code, msg = oo.verify(proofs3[k], dg, cap)
        assert code == 0, msg
    po, sto, _t, msg = oo.prove(fib_inner_proofs[0], seed=1)
    assert sto == 0, msg
    diff = np.nonzero(proofs[0] != po)[0]
    assert diff.size == 0, diff[:8]


def test_two_to_one_aggregation_of_fib64_proofs(gpu, oracle, fib_circuit, fib_inner_proofs):
    """Aggregation: one circuit verifying two fib-64 proofs (the building block of a tree over a 2048-proof batch)."""
    agg = fib_circuit.build_recursive_verifier(2)
    assert int(agg.info.num_inputs) == 2 * 19861
    inp = np.concatenate([fib_inner_proofs[0], fib_inner_proofs[1]])
    swapped_bad = inp.copy()
    swapped_bad[19861 + 300] = (int(swapped_bad[19861 + 300]) + 1) % P     # second proof corrupted
    proofs, st = agg.prove(np.stack([inp, swapped_bad]), seeds=[1, 2])
    assert st.tolist() == [0, 4]
    oo = oracle.load_circuit(agg.to_blob())
    dg, cap = agg.digest()
    code, msg = oo.verify(proofs[0], dg, cap)
    assert code == 0, msg
    print("2-to-1 aggregation circuit: 2^%d rows" % int(agg.info.degree_bits))


def test_aggregation_tree_of_four_fib64_proofs(gpu, oracle, fib_circuit, fib_inputs):
    """4 fib-64 proofs -> 2 aggregation proofs -> 1 root proof, every level proved on the GPU: the shape of the tree that
    turns a 2048-proof batch into one proof (BASELINE.json north star: "final aggregation")."""
    variants = [fib_inputs] + [gpu.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)[0] for v in (1, 2, 3)]
    leaves, st = fib_circuit.prove(np.stack(variants), seeds=[21, 22, 23, 24])
    assert st.tolist() == [0] * 4
    agg1 = fib_circuit.build_recursive_verifier(2)
    l1, st = agg1.prove(np.stack([np.concatenate([leaves[0], leaves[1]]), np.concatenate([leaves[2], leaves[3]])]),
                        seeds=[1, 2])
    assert st.tolist() == [0, 0]
    agg2 = agg1.build_recursive_verifier(2)
    root, st, tm = agg2.prove(np.concatenate([l1[0], l1[1]]), seeds=[1], timings=True)
    assert st.tolist() == [0]
    print("aggregation tree: leaf 2^%d rows -> level 1 2^%d rows -> root 2^%d rows; root proof %.1f ms"
          % (int(fib_circuit.info.degree_bits), int(agg1.info.degree_bits), int(agg2.info.degree_bits), tm.total_ms))
    o2 = oracle.load_circuit(agg2.to_blob())
    dg, cap = agg2.digest()
    code, msg = o2.verify(root[0], dg, cap)
    assert code == 0, msg
    # a root built over a corrupted level-1 proof has no witness
    bad = np.concatenate([l1[0], l1[1]])
    bad[77] = (int(bad[77]) + 1) % P
    assert agg2.prove(bad, seeds=[1])[1].tolist() == [4]
