"""An INDEPENDENT reader of upstream's binary circuit form (`CircuitData::to_bytes`), in pure Python, written from the
byte-level specification in INTEGRATION.md section 5a.1 -- not from the product's `circuit_bytes.cpp`.

It parses the bytes `p25_circuit_to_bytes` writes, checks the serializer tags and payloads of the reference's gates and
generators (whose `serialize` bodies the reference DOES hold: poseidon2_gate.rs:399-405, 529-539; arithmetic_u32.rs:287-300,
445-464; interleave_u32.rs:237-247, 340-360; uninterleave_to_u32.rs:272-283, 396-412), and REBUILDS the circuit blob of
INTEGRATION.md section 5 from them, so that a field-order error shared by the library's writer and its own reader
(which round-trip through each other) no longer goes unseen.  tests/test_circuit_bytes_cpu.py and
tests/test_gpu_circuit_bytes.py import the rebuilt blob and compare circuits, digests and proofs.
"""
import struct

P = 0xFFFFFFFF00000001
ROOT_2_32 = 1753635133440165772      # two_adic.rs:35

# serializer tags (upstream's default lists, then the reference's types)
GATE_TAGS = {0: "ArithmeticGate", 1: "ArithmeticExtensionGate", 2: "BaseSumGate", 3: "ConstantGate",
             4: "CosetInterpolationGate", 5: "ExponentiationGate", 6: "LookupGate", 7: "LookupTableGate",
             8: "MulExtensionGate", 9: "NoopGate", 10: "PoseidonMdsGate", 11: "PoseidonGate", 12: "PublicInputGate",
             13: "RandomAccessGate", 14: "ReducingExtensionGate", 15: "ReducingGate", 16: "Poseidon2Gate",
             17: "U32ArithmeticGate", 18: "U32InterleaveGate", 19: "UninterleaveToU32Gate"}
# gate tag -> (blob gate kind of INTEGRATION.md section 5, expected payload)
GATE_KIND = {9: (0, ()), 3: (1, (2,)), 12: (2, ()), 2: (3, (63,)), 18: (4, (3,)), 19: (5, (2,)), 0: (6, (20,)),
             8: (7, (13,)), 5: (8, (66,)), 17: (9, (3,)), 16: (10, ()), 1: (11, (10,)), 11: (12, ()), 13: (13, (4, 4, 2)),
             15: (14, (43,)), 14: (15, (32,)), 4: (16, None), 10: (17, ())}


def root_of_unity(log_n):
    g = ROOT_2_32
    for _ in range(32 - log_n):
        g = g * g % P
    return g


def ntt(coeffs, log_n):
    """Values of the polynomial on the subgroup <w_n>, natural order (iterative radix-2, exact integers)."""
    n = 1 << log_n
    a = list(coeffs)
    for i in range(n):
        j = int(format(i, f"0{log_n}b")[::-1], 2) if log_n else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    for s in range(1, log_n + 1):
        m, h = 1 << s, 1 << (s - 1)
        wm = root_of_unity(s)
        tw = [1] * h
        for j in range(1, h):
            tw[j] = tw[j - 1] * wm % P
        for k in range(0, n, m):
            for j in range(h):
                t = tw[j] * a[k + j + h] % P
                u = a[k + j]
                a[k + j] = (u + t) % P
                a[k + j + h] = (u - t) % P
    return a


class Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def take(self, fmt, size):
        v = struct.unpack_from(fmt, self.d, self.o)
        self.o += size
        return v

    def usize(self): return self.take("<Q", 8)[0]
    def u32(self): return self.take("<I", 4)[0]
    def u8(self): return self.take("<B", 1)[0]

    def boolean(self):
        v = self.u8()
        assert v in (0, 1), "bad bool"
        return bool(v)

    def field(self):
        v = self.usize()
        assert v < P, "non-canonical field element"
        return v

    def fields(self, k):
        v = list(self.take(f"<{k}Q", 8 * k))
        assert all(x < P for x in v), "non-canonical field element"
        return v

    def hash(self): return self.fields(4)

    def target(self):
        return ("w", self.usize(), self.usize()) if self.boolean() else ("v", self.usize())

    def vec_usize(self): return [self.usize() for _ in range(self.usize())]
    def vec_target(self): return [self.target() for _ in range(self.usize())]

    def fri_config(self):
        c = dict(rate_bits=self.usize(), cap_height=self.usize(), num_query_rounds=self.usize(), proof_of_work_bits=self.u32())
        assert self.u8() == 1, "FriReductionStrategy::ConstantArityBits expected"
        c["arity_bits"], c["final_poly_bits"] = self.usize(), self.usize()
        return c


def parse(data):
    """CircuitData bytes -> a dict of everything in them (section 5a.1, field by field).  Consumes every byte."""
    r = Reader(data)
    c = {}
    for k in ("num_wires", "num_routed_wires", "num_constants", "security_bits", "num_challenges", "max_quotient_degree_factor"):
        c[k] = r.usize()
    c["use_base_arithmetic_gate"], c["zero_knowledge"] = r.boolean(), r.boolean()
    c["fri_config"] = r.fri_config()
    c["fri_params_config"] = r.fri_config()
    c["reduction_arity_bits"] = r.vec_usize()
    c["degree_bits"] = r.usize()
    c["hiding"] = r.boolean()
    c["selector_indices"] = r.vec_usize()
    c["groups"] = [(r.usize(), r.usize()) for _ in range(r.usize())]
    for k in ("quotient_degree_factor", "num_gate_constraints", "num_constants_again", "num_public_inputs"):
        c[k] = r.usize()
    c["k_is"] = r.fields(r.usize())
    c["num_partial_products"] = r.usize()
    assert (r.usize(), r.usize(), r.usize()) == (0, 0, 0), "lookup tables"
    gates = []
    for _ in range(r.usize()):
        tag = r.u32()
        assert tag in GATE_KIND, f"gate tag {tag} ({GATE_TAGS.get(tag)}) has no evaluator in the library"
        if tag == 4:      # CosetInterpolationGate: subgroup_bits, degree, vec<F> weights
            payload = (r.usize(), r.usize(), tuple(r.fields(r.usize())))
        else:
            payload = tuple(r.usize() for _ in range(len(GATE_KIND[tag][1])))
        gates.append((tag, payload))
    c["gates"] = gates
    n, W = 1 << c["degree_bits"], c["num_wires"]
    # ---- ProverOnlyCircuitData
    gens = []
    for _ in range(r.usize()):
        tag = r.u32()
        if tag in (0, 1): pl = (r.usize(), r.field(), r.field(), r.usize())                 # row, c0, c1, i
        elif tag == 2: pl = (r.usize(), r.usize())                                          # row, num_limbs
        elif tag == 3: pl = (r.usize(), tuple(r.vec_target()))                              # row, limbs
        elif tag == 4: pl = (r.usize(), r.usize(), r.usize(), r.field())                    # row, constant_index, wire_index, constant
        elif tag == 8: pl = (r.usize(), r.usize())                                          # row, num_power_bits
        elif tag == 9: pl = (r.usize(), r.usize(), r.usize(), tuple(r.fields(r.usize())))   # row, gate payload
        elif tag == 12: pl = (r.target(), r.usize(), r.target(), r.target())                # integer, n_log, low, high
        elif tag == 13: pl = (r.usize(), r.field(), r.usize())                              # row, c0, i
        elif tag in (15, 16, 24): pl = (r.usize(),)                                         # row
        elif tag == 17: pl = tuple(r.target() for _ in range(6))
        elif tag == 18: pl = (r.usize(), r.usize(), r.usize(), r.usize(), r.usize())        # row, copy, gate payload
        elif tag == 19: pl = (r.target(),)
        elif tag in (20, 21): pl = (r.usize(), r.usize())                                   # row, num_coeffs
        elif tag == 23: pl = (r.target(), tuple(r.vec_usize()), r.usize())                  # integer, rows, num_limbs
        elif tag in (25, 26, 27): pl = (r.usize(), r.usize(), r.usize())                    # num_ops, row, i
        else: raise AssertionError(f"generator tag {tag} has no body in the library")
        gens.append((tag, pl))
    c["generators"] = gens
    c["watches"] = [(r.usize(), r.vec_usize()) for _ in range(r.usize())]
    polys = []
    for _ in range(r.usize()):
        assert r.usize() == n
        polys.append(r.fields(n))
    c["cs_coeffs"] = polys
    n_leaves = r.usize()
    assert n_leaves == n << c["fri_config"]["rate_bits"]
    c["leaves_at"] = r.o
    for _ in range(n_leaves):            # recomputed by whoever imports; kept as an offset (600 MB for the fib-64 circuit)
        w = r.usize()
        assert w == len(polys)
        r.o += 8 * w
    nd = r.usize()
    c["digests_at"], c["num_digests"] = r.o, nd
    r.o += 32 * nd
    assert r.usize() == c["fri_config"]["cap_height"]
    c["cap"] = [r.hash() for _ in range(1 << c["fri_config"]["cap_height"])]
    assert (r.usize(), r.usize(), r.boolean()) == (c["degree_bits"], c["fri_config"]["rate_bits"], False)
    assert r.usize() == n
    sig = []
    for _ in range(n):
        assert r.usize() == c["num_routed_wires"]
        sig.append(r.fields(c["num_routed_wires"]))
    c["sigmas_rows"] = sig
    assert r.usize() == n
    c["subgroup"] = r.fields(n)
    c["public_inputs"] = r.vec_target()
    c["representative_map"] = r.vec_usize()
    if r.boolean():
        c["fft_root_table"] = [r.fields(r.usize()) for _ in range(r.usize())]
    c["circuit_digest"] = r.hash()
    assert (r.usize(), r.usize()) == (0, 0), "lookups"
    # ---- VerifierOnlyCircuitData
    assert r.usize() == c["fri_config"]["cap_height"]
    c["verifier_cap"] = [r.hash() for _ in range(1 << c["fri_config"]["cap_height"])]
    c["verifier_digest"] = r.hash()
    assert r.o == len(data), f"{len(data) - r.o} trailing bytes"
    return c


def check_reference_payloads(c):
    """The tags and payloads of the reference's own gates and generators are what their `serialize` bodies write, and
    every gate generator sits on a row of its gate.  Returns {gate name: rows}."""
    n = 1 << c["degree_bits"]
    for tag, payload in c["gates"]:
        want = GATE_KIND[tag][1]
        if want is not None:
            assert payload == want, (GATE_TAGS[tag], payload, want)
        else:   # CosetInterpolationGate::with_max_degree(4, 8): weights x_i / 16 over the order-16 subgroup
            g, inv16 = root_of_unity(4), pow(16, P - 2, P)
            assert payload[:2] == (4, 6) and payload[2] == tuple(pow(g, i, P) * inv16 % P for i in range(16))
    # the gate of every row, from the selector polynomials' VALUES
    ns = len(c["groups"])
    sel = [ntt(c["cs_coeffs"][s], c["degree_bits"]) for s in range(ns)]
    row_gate = []
    for row in range(n):
        found = [v for s in range(ns) for v in [sel[s][row]] if v != 0xFFFFFFFF]
        assert len(found) == 1 and found[0] < len(c["gates"]) and c["selector_indices"][found[0]] in range(ns), row
        row_gate.append(found[0])
    c["row_gate"] = row_gate
    tag_of_row = [c["gates"][g][0] for g in row_gate]
    rows_per_tag = {}
    for t in tag_of_row:
        rows_per_tag[t] = rows_per_tag.get(t, 0) + 1
    gate_of_generator = {0: 0, 1: 1, 2: 2, 3: 2, 4: 3, 8: 5, 9: 4, 13: 8, 15: 11, 16: 10, 18: 13, 20: 15, 21: 14, 24: 16,
                         25: 17, 26: 18, 27: 19}
    num_ops = {25: 3, 26: 3, 27: 2}
    seen = set()
    for tag, pl in c["generators"]:
        if tag not in gate_of_generator:
            continue
        row = pl[1] if tag in num_ops else pl[0]
        assert row < n and tag_of_row[row] == gate_of_generator[tag], (tag, row, tag_of_row[row])
        if tag in num_ops:      # the reference's u32 generators: gate.serialize (num_ops), row, i
            assert pl[0] == num_ops[tag] and pl[2] < num_ops[tag], (tag, pl)
        if tag in (24, 25, 26, 27, 15):
            key = (tag, row, pl[2] if tag in num_ops else 0)
            assert key not in seen, key
            seen.add(key)
    # one Poseidon2Generator per Poseidon2Gate row (poseidon2_gate.rs: `generators` returns exactly one)
    assert sum(1 for t, _ in c["generators"] if t == 24) == rows_per_tag.get(16, 0)
    assert sum(1 for t, _ in c["generators"] if t == 15) == rows_per_tag.get(11, 0)
    return {GATE_TAGS[t]: k for t, k in rows_per_tag.items()}


def to_blob(c, input_target_indices):
    """The circuit blob of INTEGRATION.md section 5, rebuilt from the parsed CircuitData."""
    n, W, RW = 1 << c["degree_bits"], c["num_wires"], c["num_routed_wires"]
    if "row_gate" not in c:
        check_reference_payloads(c)

    def tidx(t):
        return t[1] * W + t[2] if t[0] == "w" else n * W + t[1]

    def wr(row, col):
        return row * W + col

    out = bytearray(b"P25CIRC1")

    def u64s(v): out.extend(struct.pack(f"<{len(v)}Q", *v))

    def u32s(v):
        out.extend(struct.pack(f"<{len(v)}I", *v))
        if len(v) % 2:
            out.extend(b"\0\0\0\0")

    kinds = [GATE_KIND[tag][0] for tag, _ in c["gates"]]
    row_kind = [kinds[g] for g in c["row_gate"]]
    values = [ntt(p, c["degree_bits"]) for p in c["cs_coeffs"]]
    ns = len(c["groups"])
    # sigmas given twice in the bytes (coefficient polynomials and the row-major table): they must agree
    for row in range(n):
        assert c["sigmas_rows"][row] == [values[len(values) - RW + j][row] for j in range(RW)], row
    gens = []
    for tag, pl in c["generators"]:
        if tag == 0: gens.append((2, pl[1], pl[2], 0, [wr(pl[0], 4 * pl[3] + k) for k in range(3)], [wr(pl[0], 4 * pl[3] + 3)]))
        elif tag == 1: gens.append((14, pl[1], pl[2], 0, [wr(pl[0], 8 * pl[3] + k) for k in range(6)], [wr(pl[0], 8 * pl[3] + 6), wr(pl[0], 8 * pl[3] + 7)]))
        elif tag == 2: gens.append((5, 0, 0, 0, [wr(pl[0], 0)], [wr(pl[0], 1 + l) for l in range(pl[1])]))
        elif tag == 3: gens.append((7, 0, 0, 0, [tidx(t) for t in pl[1]], [wr(pl[0], 0)]))
        elif tag == 4: gens.append((0, pl[3], 0, 0, [], [wr(pl[0], pl[2])]))
        elif tag == 8: gens.append((9, 0, 0, 0, [wr(pl[0], k) for k in range(pl[1] + 1)],
                                    [wr(pl[0], 2 + pl[1] + k) for k in range(pl[1])] + [wr(pl[0], 1 + pl[1])]))
        elif tag == 9: gens.append((19, 0, 0, 0, [wr(pl[0], k) for k in range(35)],
                                    [wr(pl[0], k) for k in (45, 46, 37, 38, 41, 42, 39, 40, 43, 44, 35, 36)]))
        elif tag == 12: gens.append((8, 0, 0, pl[1], [tidx(pl[0])], [tidx(pl[2]), tidx(pl[3])]))
        elif tag == 13: gens.append((3, pl[1], 0, 0, [wr(pl[0], 6 * pl[2] + k) for k in range(4)], [wr(pl[0], 6 * pl[2] + 4), wr(pl[0], 6 * pl[2] + 5)]))
        elif tag in (15, 24):
            r0 = pl[0]
            gens.append((15 if tag == 15 else 10, 0, 0, 0, [wr(r0, k) for k in range(12)] + [wr(r0, 24)],
                         [wr(r0, 25 + k) for k in range(4)] + [wr(r0, 29 + k) for k in range(106)] + [wr(r0, 12 + k) for k in range(12)]))
        elif tag == 16: gens.append((20, 0, 0, 0, [wr(pl[0], k) for k in range(24)], [wr(pl[0], 24 + k) for k in range(24)]))
        elif tag == 17: gens.append((4, 0, 0, 0, [tidx(t) for t in pl[:4]], [tidx(t) for t in pl[4:]]))
        elif tag == 18:
            r0, cp = pl[0], pl[1]
            gens.append((16, 0, 0, 0, [wr(r0, 18 * cp)] + [wr(r0, 18 * cp + 2 + k) for k in range(16)],
                         [wr(r0, 18 * cp + 1)] + [wr(r0, 74 + 4 * cp + k) for k in range(4)]))
        elif tag == 19: gens.append((1, 0, 0, pl[0][2], [], [tidx(pl[0])]))
        elif tag in (20, 21):
            r0, N, w = pl[0], pl[1], (1 if tag == 20 else 2)
            outs = []
            for k in range(N):
                w0 = 0 if k == N - 1 else 6 + N * w + 2 * k
                outs += [wr(r0, w0), wr(r0, w0 + 1)]
            gens.append((17 if tag == 20 else 18, 0, 0, 0, [wr(r0, k) for k in range(2, 6 + N * w)], outs))
        elif tag == 23: gens.append((6, 0, 0, 0, [tidx(pl[0])], [wr(rw, 0) for rw in pl[1]]))
        elif tag == 25: gens.append((11, 0, 0, 0, [wr(pl[1], 6 * pl[2] + k) for k in range(3)],
                                     [wr(pl[1], 6 * pl[2] + 3 + k) for k in range(3)] + [wr(pl[1], 18 + 32 * pl[2] + j) for j in range(32)]))
        elif tag == 26: gens.append((12, 0, 0, 0, [wr(pl[1], 2 * pl[2])],
                                     [wr(pl[1], 6 + 32 * pl[2] + j) for j in range(32)] + [wr(pl[1], 2 * pl[2] + 1)]))
        elif tag == 27: gens.append((13, 0, 0, 0, [wr(pl[1], 3 * pl[2])],
                                     [wr(pl[1], 6 + 64 * pl[2] + j) for j in range(64)] + [wr(pl[1], 3 * pl[2] + 1), wr(pl[1], 3 * pl[2] + 2)]))
        else:
            raise AssertionError(tag)
    pi_rows = [row for row in range(n) if row_kind[row] == 2]
    header = [0] * 32
    header[0] = c["degree_bits"]
    header[1], header[2], header[3] = W, RW, c["num_constants"]
    header[4], header[5] = c["num_challenges"], c["max_quotient_degree_factor"]
    header[6], header[7] = c["fri_config"]["rate_bits"], c["fri_config"]["cap_height"]
    header[8], header[9] = c["fri_config"]["proof_of_work_bits"], c["fri_config"]["num_query_rounds"]
    header[10] = len(c["reduction_arity_bits"])
    header[11] = ns
    header[12] = c["num_gate_constraints"]
    header[13] = c["num_partial_products"]
    header[14] = len(kinds)
    header[15] = pi_rows[0] if pi_rows else (1 << 64) - 1
    header[16] = len(c["representative_map"]) - n * W
    header[17] = len(input_target_indices)
    header[18] = len(gens)
    header[19] = len(values)
    header[20], header[21] = c["fri_config"]["arity_bits"], c["fri_config"]["final_poly_bits"]
    header[22] = len(c["public_inputs"])
    u64s(header)
    for i, k in enumerate(kinds):
        s = c["selector_indices"][i]
        u64s([k, s, c["groups"][s][0], c["groups"][s][1]])
    u64s(c["reduction_arity_bits"])
    u32s(row_kind)
    for v in values:
        u64s(v)
    u64s(c["k_is"])
    u32s([int(t) for t in input_target_indices])
    u32s(c["representative_map"])
    for kind, c0, c1, aux, deps, outs in gens:
        u64s([kind, c0, c1, aux, len(deps), len(outs)])
        u32s(deps + outs)
    if c["public_inputs"]:
        u32s([tidx(t) for t in c["public_inputs"]])
    return bytes(out)
