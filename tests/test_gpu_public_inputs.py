"""GPU: public inputs on the device path -- values gathered from the witness, hashed by one wave per proof, fed to the
transcript and to the PublicInputGate evaluator of the quotient kernel, appended to the flat proof -- byte-equal to
the oracle's proofs; aggregation circuits exposing a commitment to the batch they verify."""
import numpy as np
import pytest

from conftest import P

pytestmark = pytest.mark.gpu


def test_public_input_proofs_equal_the_oracles(gpu, oracle):
    n = 9                                                     # 17 public inputs: three chunks of the rate-8 sponge
    c = gpu.Circuit.build_gadget(11, n)
    oc = oracle.load_circuit(c.to_blob())
    rng = np.random.default_rng(7)
    batch = (rng.integers(0, 1 << 62, size=(5, n), dtype=np.uint64) * np.uint64(3)) % np.uint64(P)
    proofs, st = c.prove(batch, seeds=[1, 2, 3, 4, 5])
    assert st.tolist() == [0] * 5
    dg, cap = c.digest()
    for i in range(5):
        po, sto, _t, msg = oc.prove(batch[i], seed=i + 1)
        assert sto == 0, msg
        assert (proofs[i] == po).all(), f"proof {i} differs from the oracle's"
        pis = c.public_inputs(proofs[i])
        prod = 1
        for v in batch[i]:
            prod = prod * int(v) % P
        assert (pis[:n] == batch[i]).all() and int(pis[-1]) == prod
        assert oc.verify(proofs[i], dg, cap)[0] == 0
    # one proof alone (latency path) gives the same bytes
    single, st1 = c.prove(batch[2], seeds=[3])
    assert st1.tolist() == [0] and (single[0] == proofs[2]).all()
    # the stage entry point takes the hash from the witness's PublicInputGate row
    wires, stw = c.witness(batch[0], seed=1)
    assert stw == 0
    betas, gammas, alphas = ([11, 12], [13, 14], [15, 16])
    zs = c.partial_products(wires, betas, gammas)
    assert (c.quotient(wires, zs, betas, gammas, alphas) == oc.quotient(wires, zs, betas, gammas, alphas)).all()


def test_aggregation_tree_commits_to_its_leaves(gpu, oracle):
    """4 leaf proofs (a plonky3-verifier circuit: no public inputs) -> 2 aggregates -> 1 root, all on the GPU; the
    root's four public inputs are the Poseidon tree over hash_no_pad(leaf wires cap)."""
    inp, cfg = gpu.p3_prove_fibonacci(3, 3, 4)
    leaf = gpu.Circuit.build_p3_verifier(cfg)
    leaves, st = leaf.prove(np.stack([inp] * 4), seeds=[1, 2, 3, 4])
    assert st.tolist() == [0] * 4
    ids = [oracle.hash_no_pad(p[:64]) for p in leaves]
    a1 = leaf.build_aggregator(2)
    mid, st = a1.prove(np.stack([np.concatenate([leaves[0], leaves[1]]), np.concatenate([leaves[2], leaves[3]])]), seeds=[5, 6])
    assert st.tolist() == [0, 0]
    m = [oracle.hash_no_pad(np.concatenate([ids[0], ids[1]])), oracle.hash_no_pad(np.concatenate([ids[2], ids[3]]))]
    assert (a1.public_inputs(mid[0]) == m[0]).all() and (a1.public_inputs(mid[1]) == m[1]).all()
    a2 = a1.build_aggregator(2)
    root, st = a2.prove(np.concatenate([mid[0], mid[1]])[None, :], seeds=[7])
    assert st.tolist() == [0]
    assert (a2.public_inputs(root[0]) == oracle.hash_no_pad(np.concatenate(m))).all()
    oc = oracle.load_circuit(a2.to_blob())
    dg, cap = a2.digest()
    assert oc.verify(root[0], dg, cap)[0] == 0
    po, sto, _t, msg = oc.prove(np.concatenate([mid[0], mid[1]]), seed=7)
    assert sto == 0 and (po == root[0]).all()
    # a leaf proof swapped for another valid one changes the commitment; a corrupted one has no witness
    other, st = leaf.prove(inp[None, :], seeds=[99])
    mid2, st = a1.prove(np.concatenate([other[0], leaves[1]])[None, :], seeds=[5])
    assert st.tolist() == [0] and (a1.public_inputs(mid2[0]) != m[0]).any()
    bad = np.concatenate([leaves[0], leaves[1]])
    bad[70] = (int(bad[70]) + 1) % P
    _p, st = a1.prove(bad[None, :], seeds=[5])
    assert st.tolist() == [4]
