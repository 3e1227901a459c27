"""Gadget circuits mirroring the reference's gadget tests, with natively computed expectations
(the reference tests compare against native u64 ops: src/p3/mod.rs:271-494)."""
import numpy as np

P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1
TWO_ADIC = 1753635133440165772


def rev_bits(x, n):
    return int(format(x, "064b")[::-1], 2) >> (64 - n)


def cases(oracle=None):
    rng = np.random.default_rng(20240611)

    def r64():
        return int(rng.integers(0, P, dtype=np.uint64))  # canonical field element used as a u64

    out = []
    x, y = r64(), r64()
    out.append(("and", 0, 0, [x, y, (x & y) % P]))
    out.append(("xor", 1, 0, [x, y, (x ^ y) % P]))
    for n in (1, 7, 31, 33, 63):
        x = r64()
        out.append((f"lsh{n}", 2, n, [x, ((x << n) & M64) % P]))
        out.append((f"rsh{n}", 3, n, [x, (x >> n) % P]))
    for n in (7, 19, 64):
        x = r64()
        out.append((f"rev{n}", 4, n, [x, rev_bits(x, n) % P]))
    e = int(rng.integers(0, 1 << 19))
    w = pow(TWO_ADIC, 1 << (32 - 19), P)
    out.append(("exp19", 6, 19, [e, 7 * pow(w, e, P) % P]))
    a = r64()
    out.append(("connected_inputs", 8, 0, [a, a, a * a % P]))
    # extension-field chain (gadget 9): F_p[X]/(X^2 - 7)
    def emul(x, y):
        return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)

    def eadd(x, y):
        return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)

    def einv(x):
        n = pow((x[0] * x[0] - 7 * x[1] * x[1]) % P, P - 2, P)
        return (x[0] * n % P, (P - x[1]) * n % P)

    ea, eb, ec = (r64(), r64()), (r64(), r64()), (r64(), r64())
    t = eadd(emul(ea, eb), ec)
    t = eadd(emul(t, ea), ((P - eb[0]) % P, (P - eb[1]) % P))
    t = eadd((5 * t[0] % P, 5 * t[1] % P), (3, 9))
    t7 = t
    for _ in range(6):
        t7 = emul(t7, t)
    q = emul(t7, einv(ec))
    s = eadd(eadd(eadd(q, ea), eb), (ea[0] * eb[0] % P, ea[0] * eb[1] % P))
    out.append(("ext_arith", 9, 0, list(ea) + list(eb) + list(ec) + list(s)))
    if oracle is not None:
        # Poseidon v1 hashing + a Merkle step (gadget 10) for leaf widths on both sides of the noop / rate boundaries
        for width in (3, 4, 5, 8, 9, 20, 135):
            for bit in (0, 1):
                leaf = [r64() for _ in range(width)]
                sib = [r64() for _ in range(4)]
                h = leaf + [0] * (4 - width) if width <= 4 else [int(v) for v in oracle.hash_no_pad(np.array(leaf, dtype=np.uint64))]
                st = (sib + h if bit else h + sib) + [0] * 4
                parent = oracle.poseidon_permute(np.array(st, dtype=np.uint64))[0][:4]
                out.append((f"poseidon_merkle{width}_{bit}", 10, width, leaf + sib + [bit] + [int(v) for v in parent]))
        l = [r64() for _ in range(4)]
        r = [r64() for _ in range(4)]
        st = oracle.poseidon2_permute(np.array(l + r + [0] * 4, dtype=np.uint64))[0]
        out.append(("compress", 5, 0, l + r + [int(v) for v in st[:4]]))
        # hash_iter_slices: Poseidon2 overwrite-mode sponge, RATE = 4 (src/p3/constants.rs:3), over the
        # concatenated slices
        for n_slices in (2, 3, 5):   # 2 = the reference's test_hash_iter_slice shape
            words = [r64() for _ in range(4 * n_slices)]
            state = np.zeros(12, dtype=np.uint64)
            for off in range(0, len(words), 4):
                chunk = words[off:off + 4]
                state[:len(chunk)] = chunk
                state = oracle.poseidon2_permute(state)[0].copy()
            out.append((f"hash_slices{n_slices}", 7, n_slices, words + [int(v) for v in state[:4]]))
    return out
