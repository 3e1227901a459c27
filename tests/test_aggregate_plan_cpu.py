"""plonky25_amd.aggregate's tree shapes (host logic, no GPU): how a level of n children is cut into groups when the
arity does not divide it, and that the commitment recomputed from the leaves follows the same shape."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


@pytest.fixture(scope="module")
def ag():
    ge.load_package()
    from plonky25_amd import aggregate
    return aggregate


def test_powers_of_two_fold_as_before(ag):
    assert ag.level_plan(256, 8) == [8, 8, 4]
    assert ag.level_plan(64, 8) == [8, 8]
    assert ag.level_plan(4, 8) == [4]
    assert ag.level_plan(8, 2) == [2, 2, 2]
    assert ag.level_plan(1, 8) == []
    assert ag.group_bounds(32, 8) == [(0, 8), (8, 16), (16, 24), (24, 32)]


def test_the_bench_shape_256_by_13(ag):
    # 20 groups of 13 over 256 leaves (the last one rows 243..255), 2 groups of 10, 1 of 2: 23 aggregates, not 37
    assert ag.level_plan(256, 13) == [13, 10, 2]
    b = ag.group_bounds(256, 13)
    assert len(b) == 20 and b[0] == (0, 13) and b[18] == (234, 247) and b[19] == (243, 256)
    assert ag.group_bounds(20, 10) == [(0, 10), (10, 20)]


@pytest.mark.parametrize("n", list(range(1, 70)) + [255, 256, 257, 1000])
@pytest.mark.parametrize("arity", [2, 3, 5, 8, 13, 16])
def test_every_child_is_in_a_group_and_groups_fit(ag, n, arity):
    plan, m = ag.level_plan(n, arity), n
    for k in plan:
        assert 2 <= k <= arity or (k == m and k <= arity)
        bounds = ag.group_bounds(m, k)
        assert len(bounds) == -(-m // k) == -(-m // arity)          # as few groups as the arity allows
        covered = set()
        for a, b in bounds:
            assert 0 <= a < b <= m and b - a == k
            covered.update(range(a, b))
        assert covered == set(range(m))
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(len(bounds) - 2))   # only the last one may overlap
        m = len(bounds)
    assert m == 1


def test_expected_commitment_follows_the_plan(ag):
    def h(words):   # order- and length-sensitive stand-in for hash_no_pad
        w = np.asarray(words, dtype=np.uint64)
        acc = np.uint64(len(w) + 1)
        out = []
        for j in range(4):
            for i, v in enumerate(w):
                acc = np.uint64((int(acc) * 6364136223846793005 + int(v) + i + j) % (1 << 64))
            out.append(acc)
        return np.array(out, dtype=np.uint64)

    rng = np.random.default_rng(3)
    caps = [rng.integers(0, 1 << 62, size=64, dtype=np.uint64) for _ in range(5)]
    ids = [h(c) for c in caps]
    l1 = [h(np.concatenate(ids[0:3])), h(np.concatenate(ids[2:5]))]
    want = [int(v) for v in h(np.concatenate(l1))]
    assert ag.expected_commitment(caps, 3, h) == want
    # two shards of five: each folded by itself, the two roots hashed once more
    caps2 = caps + [rng.integers(0, 1 << 62, size=64, dtype=np.uint64) for _ in range(5)]
    ids2 = [h(c) for c in caps2[5:]]
    r2 = h(np.concatenate([h(np.concatenate(ids2[0:3])), h(np.concatenate(ids2[2:5]))]))
    assert ag.expected_commitment(caps2, 3, h, n_shards=2) == [int(v) for v in h(np.concatenate([h(np.concatenate(l1)), r2]))]


def test_widest_arity_fills_a_power_of_two(ag):
    """rows(k) = a + b k from two small builds; the arity with the fewest padded rows per child; the chosen circuit is
    built and checked, one child fewer when the estimate was short."""
    class Fake:
        built = []

        def __init__(self, a, b, k=0, bump=0):
            self.a, self.b, self.bump = a, b, bump
            rows = a + b * k + (bump if k > 3 else 0)

            class Info:
                num_rows_used = rows
                degree_bits = max(1, (rows - 1).bit_length())
            self.info = Info()

        def build_aggregator(self, k):
            Fake.built.append(k)
            return Fake(self.a, self.b, k, self.bump)

        def close(self):
            pass

    assert ag.widest_arity(Fake(181, 4813)) == 13            # the fib-64 verifier circuit's figures: 62,750 of 2^16 rows
    assert Fake.built[:2] == [2, 3] and Fake.built[-1] == 13
    assert ag.widest_arity(Fake(181, 4813), cap=8) == 6       # 2^15 / 6 beats 2^16 / 8
    assert ag.widest_arity(Fake(100, 1000)) == 16             # 16,100 rows in 2^14: the cap
    assert ag.widest_arity(Fake(181, 4813, bump=3000)) == 12  # the real circuit larger than the line through k = 2, 3
