"""GPU: recursion -- circuits that verify this library's proofs, proved on the MI355X and checked
against the oracle: gate-level evaluator circuits, the recursive verifier of a small circuit, of a fib-64
plonky3-verifier proof (the batch item of BASELINE.json), and a 2-to-1 aggregation of two such proofs."""
import numpy as np
import pytest

from conftest import P, splitmix_field

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", list(range(1, 18)))
def test_gate_eval_circuits_gpu_equals_oracle(gpu, oracle, kind):
    """The in-circuit evaluator of EVERY gate kind (1-10 the inner circuits', 11-17 what recursion adds), as a circuit
    proved on the GPU: byte-equal to the oracle's proof, a wrong expectation has no witness."""
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


@pytest.fixture(scope="module")
def rec_small(gpu, oracle):
    """Depth-2 recursion over the and(x, y) gadget circuit: the verifier (2^12 rows) of a recursive verifier (2^11
    rows).  It holds EVERY gate recursion adds -- ArithmeticExtension, Poseidon, RandomAccess, Reducing,
    ReducingExtension, CosetInterpolation (the inner proof has FRI layers) and PoseidonMds (the inner circuit has
    PoseidonGate rows, whose in-circuit evaluator puts its MDS layers on such rows)."""
    inner = gpu.Circuit.build_gadget(0, 0)
    x, y = 0x0123456789ABCDEF % P, 0x0FEDCBA987654321 % P
    inp = np.array([x, y, (x & y) % P], dtype=np.uint64)
    inner_proofs, st = inner.prove(inp[None, :], seeds=[5])
    assert st.tolist() == [0]
    outer1 = inner.build_recursive_verifier(1)
    p1, st = outer1.prove(inner_proofs[:1], seeds=[6])
    assert st.tolist() == [0]
    outer = outer1.build_recursive_verifier(1)
    oo = oracle.load_circuit(outer.to_blob())
    wires, st, msg = oo.witness(p1[0], seed=9)
    assert st == 0, msg
    return outer, oo, wires


def _chal(seed):
    return splitmix_field(6, seed=seed).reshape(3, 2)


def _row_kinds(blob):
    """Gate kind of every row, from the circuit blob (magic, 32 header words, the gate table
    of header[14] entries x 4 words, header[10] FRI arity words, then u32[n])."""
    hdr = np.frombuffer(blob, dtype=np.uint64, count=32, offset=8)
    off = 8 + 32 * 8 + int(hdr[14]) * 32 + int(hdr[10]) * 8
    return np.frombuffer(blob, dtype=np.uint32, count=1 << int(hdr[0]), offset=off).copy()


def test_quotient_rec_evaluators_vs_oracle_on_arbitrary_wires(rec_small):
    """k_quotient_rec DIRECTLY (p25_quotient, the stage entry point) against the oracle's ref_quotient_chunks on a
    recursion circuit: on the satisfied witness, and on ARBITRARY wires -- every row's evaluator then produces non-zero
    constraints from every wire it reads, so each of the recursion gates' evaluators (kinds 11-17) is compared value
    for value, not only through a proof whose constraints vanish on the subgroup."""
    outer, oo, wires = rec_small
    counts = outer.gate_counts()
    present = [name for name, k in counts.items() if k]
    for need in ("ArithmeticExtensionGate", "PoseidonGate(", "RandomAccessGate", "ReducingGate", "ReducingExtensionGate",
                 "CosetInterpolationGate", "PoseidonMdsGate"):
        assert any(need in name for name in present), (need, present)
    betas, gammas, alphas = _chal(31)
    zs = oo.partial_products(wires, betas, gammas)
    assert (outer.partial_products(wires, betas, gammas) == zs).all()
    qg, qo = outer.quotient(wires, zs, betas, gammas, alphas), oo.quotient(wires, zs, betas, gammas, alphas)
    assert qg.shape == qo.shape and (qg == qo).all() and qg.any()
    # arbitrary wires (canonical field elements everywhere): same words on both sides
    rnd = splitmix_field(wires.size, seed=4242).reshape(wires.shape)
    zs_r = oo.partial_products(rnd, betas, gammas)
    qg, qo = outer.quotient(rnd, zs_r, betas, gammas, alphas), oo.quotient(rnd, zs_r, betas, gammas, alphas)
    assert (qg == qo).all(), np.argwhere(qg != qo)[:5]
    # and a proof on the same context afterwards is still the oracle's (the stage call leaves no state behind)
    assert (qg != outer.quotient(wires, zs, betas, gammas, alphas)).any()


def test_quotient_rec_sees_every_wire_of_every_recursion_gate(rec_small):
    """One wire column at a time, on the first row of each gate type of the recursion circuit: changing it changes the
    GPU's quotient exactly as it changes the oracle's."""
    outer, oo, wires = rec_small
    betas, gammas, alphas = _chal(32)
    zs = oo.partial_products(wires, betas, gammas)
    base = oo.quotient(wires, zs, betas, gammas, alphas)
    kinds = _row_kinds(outer.to_blob())
    assert set(range(11, 18)) <= set(np.unique(kinds).tolist())
    rows = sorted({int(np.nonzero(kinds == k)[0][0]) for k in np.unique(kinds) if k != 0})
    rng = np.random.default_rng(77)
    for row in rows:
        for col in rng.choice(wires.shape[0], size=6, replace=False):
            bad = wires.copy()
            bad[col, row] = (int(bad[col, row]) + 1) % P
            qg = outer.quotient(bad, zs, betas, gammas, alphas)
            qo = oo.quotient(bad, zs, betas, gammas, alphas)
            assert (qg == qo).all(), (row, col)
    assert base.any()


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


def test_device_tree_chained_on_the_device_equals_the_host_fold(gpu, oracle):
    """plonky25_amd.aggregate.DeviceTree -- aggregation levels proving straight on the buffer the level below writes,
    ordered by p25_circuit_mark / p25_circuit_wait_mark, level l lagged l steps, nothing synchronised in between -- gives,
    for every step, byte for byte the root the host-side fold of the same leaves gives (same circuits, same seeds), and
    the root commits to the leaves.  Seven steps through a two-level tree with four buffer slots: slots are reused."""
    import torch
    from plonky25_amd import aggregate as ag
    leaf = gpu.Circuit.build_gadget(0, 0)                      # and(x, y): 2^4 rows; its aggregators: 2^12, 2^13 rows
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    n_leaves, steps = 4, 7          # slots = 4: steps 4..6 reuse buffers, so the reader-has-finished waits execute
    pw = int(leaf.info.proof_words)
    inputs = []
    for s in range(steps):
        xs = rng.integers(0, P, size=(n_leaves, 2), dtype=np.uint64)
        inputs.append(np.stack([np.array([int(a), int(b), (int(a) & int(b)) % P], dtype=np.uint64) for a, b in xs]))
    tree = ag.DeviceTree(leaf, n_leaves, 2, dev, level_streams=(2, 1))
    assert [L["n"] for L in tree.levels] == [2, 1] and tree.slots == 4
    d_in = [torch.from_numpy(x.view(np.int64)).to(dev) for x in inputs]
    d_seeds = torch.arange(n_leaves, dtype=torch.int64, device=dev)
    d_st = torch.zeros((steps, n_leaves), dtype=torch.int32, device=dev)

    def leaves(buf, j):
        leaf.prove_dev(d_in[j].data_ptr(), n_leaves, d_seeds.data_ptr(), buf.data_ptr(), pw, d_st[j].data_ptr())

    roots = {}
    for j in range(steps):
        tree.step(leaves)
    tree.flush()
    tree.sync()
    torch.cuda.synchronize()
    assert int((d_st != 0).sum().item()) == 0
    for j in range(steps - tree.slots, steps):
        if j >= 0:
            roots[j] = tree.root(j)
    # the host-side fold of the same leaves: same aggregation circuits, same seeds -> the same bytes
    for j, (root, ok) in roots.items():
        assert ok, j
        lp, st = leaf.prove(inputs[j], seeds=np.arange(n_leaves, dtype=np.uint64))
        assert st.tolist() == [0] * n_leaves
        assert (tree.leaf_proofs(j).cpu().numpy().view(np.uint64)[:n_leaves] == lp).all(), j
        f = ag.fold(leaf, [lp[i] for i in range(n_leaves)], arity=2, warm=False)
        assert (f["root"] == root).all(), f"step {j}: the device-chained root differs from the host fold's"
        want = ag.expected_commitment([lp[i][:ag.CAP_WORDS] for i in range(n_leaves)], 2, oracle.hash_no_pad)
        assert [int(v) for v in tree.top.public_inputs(root)] == want
        oc = oracle.load_circuit(f["top"].to_blob())
        dg, cap = f["top"].digest()
        assert oc.verify(root, dg, cap)[0] == 0
        for c in f["owned"]:
            c.close()
    tree.close()


def test_device_tree_and_fold_with_an_arity_that_does_not_divide_the_leaves(gpu, oracle):
    """Five leaves, at most three children per aggregator: level 1 is two groups, rows 0..2 and -- right-aligned,
    overlapping -- rows 2..4 (aggregate.group_bounds), level 2 the two of them.  The device-chained tree (a second
    p25_prove_batch_dev on the last three rows of the same buffer), the host fold and the commitment recomputed from the
    leaves agree, the oracle's verifier accepts the root, and the root commits to EVERY leaf: changing any one leaf's
    cap changes the expected commitment."""
    import torch
    from plonky25_amd import aggregate as ag
    assert ag.level_plan(5, 3) == [3, 2] and ag.group_bounds(5, 3) == [(0, 3), (2, 5)]
    leaf = gpu.Circuit.build_gadget(0, 0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n_leaves, steps = 5, 5
    pw = int(leaf.info.proof_words)
    inputs = []
    for s in range(steps):
        xs = rng.integers(0, P, size=(n_leaves, 2), dtype=np.uint64)
        inputs.append(np.stack([np.array([int(a), int(b), (int(a) & int(b)) % P], dtype=np.uint64) for a, b in xs]))
    tree = ag.DeviceTree(leaf, n_leaves, 3, dev)
    assert [(L["k"], L["children"], L["n"]) for L in tree.levels] == [(3, 5, 2), (2, 2, 1)]
    d_in = [torch.from_numpy(x.view(np.int64)).to(dev) for x in inputs]
    d_seeds = torch.arange(n_leaves, dtype=torch.int64, device=dev)
    d_st = torch.zeros((steps, n_leaves), dtype=torch.int32, device=dev)

    def leaves(buf, j):
        leaf.prove_dev(d_in[j].data_ptr(), n_leaves, d_seeds.data_ptr(), buf.data_ptr(), pw, d_st[j].data_ptr())

    for j in range(steps):
        tree.step(leaves)
    tree.flush(); tree.sync(); torch.cuda.synchronize()
    assert int((d_st != 0).sum().item()) == 0
    for j in range(steps - tree.slots, steps):
        if j < 0:
            continue
        root, ok = tree.root(j)
        assert ok, j
        lp, st = leaf.prove(inputs[j], seeds=np.arange(n_leaves, dtype=np.uint64))
        assert st.tolist() == [0] * n_leaves
        f = ag.fold(leaf, [lp[i] for i in range(n_leaves)], arity=3, warm=False)
        assert [l["arity"] for l in f["levels"]] == [3, 2] and [l["proofs"] for l in f["levels"]] == [2, 1]
        assert (f["root"] == root).all(), f"step {j}: the device-chained root differs from the host fold's"
        caps = [lp[i][:ag.CAP_WORDS] for i in range(n_leaves)]
        want = ag.expected_commitment(caps, 3, oracle.hash_no_pad)
        assert [int(v) for v in tree.top.public_inputs(root)] == want
        for i in range(n_leaves):                 # every leaf is under the root
            other = [c.copy() for c in caps]
            other[i][0] = (int(other[i][0]) + 1) % P
            assert ag.expected_commitment(other, 3, oracle.hash_no_pad) != want, i
        oc = oracle.load_circuit(f["top"].to_blob())
        dg, cap = f["top"].digest()
        assert oc.verify(root, dg, cap)[0] == 0
        for c in f["owned"]:
            c.close()
    tree.close()


def test_widest_arity_of_the_fib64_verifier_circuit(gpu, fib_circuit):
    """bench.py's default (--aggregate-arity 0): thirteen fib-64 verifier proofs are what the 2^16 rows of an aggregation
    circuit hold (profiles/r04_arity.txt); fourteen need 2^17."""
    from plonky25_amd import aggregate as ag
    assert ag.widest_arity(fib_circuit) == 13
    a13, a14 = fib_circuit.build_aggregator(13), fib_circuit.build_aggregator(14)
    assert int(a13.info.degree_bits) == 16 and int(a14.info.degree_bits) == 17
    assert int(a13.info.num_inputs) == 13 * int(fib_circuit.info.proof_words)
    a13.close(); a14.close()


def test_device_tree_a_failed_leaf_fails_its_branch_only(gpu):
    """A leaf whose witness does not exist (P25_ERR_WITNESS_CONFLICT) leaves no valid proof in the buffer: the aggregate
    above it must fail too (its status says so -- never a root that looks valid), while the other steps in flight through
    the same buffers are untouched."""
    import torch
    from plonky25_amd import aggregate as ag
    leaf = gpu.Circuit.build_gadget(0, 0)
    dev = torch.device("cuda", 0)
    n_leaves, steps, bad_step = 4, 3, 1
    pw = int(leaf.info.proof_words)
    rng = np.random.default_rng(7)
    d_in = []
    for s in range(steps):
        xs = rng.integers(0, P, size=(n_leaves, 2), dtype=np.uint64)
        rows = np.stack([np.array([int(a), int(b), (int(a) & int(b)) % P], dtype=np.uint64) for a, b in xs])
        if s == bad_step:
            rows[2, 2] = (int(rows[2, 2]) + 1) % P          # and(x, y) != expected: no witness for leaf 2 of this step
        d_in.append(torch.from_numpy(rows.view(np.int64)).to(dev))
    tree = ag.DeviceTree(leaf, n_leaves, 2, dev, level_streams=(2, 1))
    d_seeds = torch.arange(n_leaves, dtype=torch.int64, device=dev)
    d_st = torch.zeros((steps, n_leaves), dtype=torch.int32, device=dev)

    def leaves(buf, j):
        leaf.prove_dev(d_in[j].data_ptr(), n_leaves, d_seeds.data_ptr(), buf.data_ptr(), pw, d_st[j].data_ptr())

    for _ in range(steps):
        tree.step(leaves)
    tree.flush()
    tree.sync()
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert st[bad_step].tolist() == [0, 0, 4, 0] and (st[[0, 2]] == 0).all()
    for j in range(steps):
        _root, ok = tree.root(j)
        assert ok == (j != bad_step), j
    l1 = tree.levels[0]["status"][bad_step % tree.slots].cpu().numpy()
    assert l1[0] == 0 and l1[1] != 0                       # only the aggregate over leaves 2, 3 failed at level 1
    tree.close()


def test_cross_rank_aggregate_at_the_shapes_of_2_4_and_8_gpus(gpu, oracle, fib_circuit, fib_inputs):
    """BASELINE config 4's final aggregation at N = 2, 4, 8 on the one GPU there is, over N DISTINCT shard roots: eight
    different shards of 64 fib-64 leaves (other plonky3 proofs in another rotation, other filler seeds: every leaf of the 512
    is a different proof) each fold to their own shard root (13 at a time: 64 -> 5 -> 1, bench.py's default; the shards share
    the level circuits, as the ranks of one job build the same ones), then rank 0's step -- `fold_roots(top, roots[:N])`, the
    N-to-1 aggregate over proofs of the shard trees' top circuit -- is built and proved for N = 2, 4, 8.  The oracle's
    verifier accepts each root and its public inputs are the commitment to the N x 64 leaves in global order
    (`expected_commitment(..., n_shards=N)`); swapping two shard roots changes the commitment."""
    from plonky25_amd import aggregate as ag
    n_leaves, arity, n_shards = 64, 13, 8
    variants = [fib_inputs] + [gpu.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)[0] for v in (1, 2, 3, 4)]
    folds, caps, level_circs = [], [], None
    for q in range(n_shards):
        batch = np.stack([variants[(q * n_leaves + i) % 5] for i in range(n_leaves)])
        leaves, st = fib_circuit.prove(batch, seeds=np.arange(n_leaves, dtype=np.uint64) + np.uint64(500 + 1000 * q))
        assert (st == 0).all()
        f = ag.fold(fib_circuit, [leaves[i] for i in range(n_leaves)], arity=arity, warm=False, circuits=level_circs)
        if level_circs is None:
            level_circs = f["owned"]
            assert [l["arity"] for l in f["levels"]] == [13, 5] and [l["proofs"] for l in f["levels"]] == [5, 1]
        shard_caps = [leaves[i][:ag.CAP_WORDS] for i in range(n_leaves)]
        assert [int(v) for v in f["top"].public_inputs(f["root"])] == ag.expected_commitment(shard_caps, arity, oracle.hash_no_pad)
        folds.append(f)
        caps.extend(shard_caps)
    top, roots = folds[0]["top"], [f["root"] for f in folds]
    assert len({r.tobytes() for r in roots}) == n_shards and len({c.tobytes() for c in caps}) == n_shards * n_leaves
    shapes = {}
    for n in (2, 4, 8):
        fin = ag.fold_roots(top, roots[:n], warm=False)
        rec = fin["levels"][0]
        assert rec["level"] == "cross-rank" and rec["arity"] == n and rec["proofs"] == 1
        shapes[n] = (rec["rows_used"], rec["circuit_rows_log2"], rec["ms_per_proof"])
        got = [int(v) for v in fin["top"].public_inputs(fin["root"])]
        assert got == ag.expected_commitment(caps[:n * n_leaves], arity, oracle.hash_no_pad, n_shards=n), n
        oc = oracle.load_circuit(fin["top"].to_blob())
        dg, cap = fin["top"].digest()
        code, msg = oc.verify(fin["root"], dg, cap)
        assert code == 0, (n, msg)
        # rank order matters: the same roots with two shards swapped commit to another batch
        swapped = [roots[1], roots[0]] + roots[2:n]
        out, st = fin["top"].prove(np.concatenate(swapped)[None, :], seeds=[0])
        assert st.tolist() == [0] and [int(v) for v in fin["top"].public_inputs(out[0])] != got
        # a shard root that is not a valid proof has no cross-rank aggregate
        bad = np.concatenate(roots[:n])
        bad[100] = (int(bad[100]) + 1) % P
        assert fin["top"].prove(bad[None, :], seeds=[0])[1].tolist() != [0]
        for c in fin["owned"]:
            c.close()
    print("cross-rank aggregation circuits (rows used, log2 rows, ms per proof, cold):", shapes)
    for c in level_circs:
        c.close()
