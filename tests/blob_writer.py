"""An INDEPENDENT writer of the circuit blob (`p25_circuit_import`), in pure Python, following the field-by-field
specification in INTEGRATION.md section 5 -- not the product's `p25_circuit_export`.

It plays the part of the Rust host that keeps `builder.build::<C>()` (/root/reference/src/p3/mod.rs:250) and hands the
resulting `CircuitData` to the library: a miniature circuit builder (arithmetic gates, constants, the public-input
row, copy constraints) that derives selectors, sigma polynomials, k_i's, the representative map and the generator
table by itself and serialises them.  tests/test_blob_independent_writer.py imports its output and proves with it.
"""
import struct

P = 0xFFFFFFFF00000001
MAGIC = b"P25CIRC1"

# gate kinds / generator kinds of the blob (INTEGRATION.md section 5, tables "gate kind" and "generator kind")
G_NOOP, G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC = 0, 1, 2, 6
G_U32_INTERLEAVE, G_U32_UNINTERLEAVE, G_U32_ARITHMETIC, G_POSEIDON2 = 4, 5, 9, 10     # the reference's own four gates
GEN_CONSTANT, GEN_RANDOM, GEN_ARITHMETIC = 0, 1, 2
GEN_POSEIDON2, GEN_U32_ARITHMETIC, GEN_U32_INTERLEAVE, GEN_U32_UNINTERLEAVE = 10, 11, 12, 13
_PHANTOM = "PhantomData<plonky2_field::goldilocks_field::GoldilocksField>"
# (degree, Gate::id()) -- orders the gate set; the reference's ids are `format!("{self:?}")` of its structs
# (interleave_u32.rs:98-100, uninterleave_to_u32.rs:110-112, arithmetic_u32.rs:102-104, poseidon2_gate.rs:146-148)
GATE_ORDER = {
    G_NOOP: (0, "NoopGate"),
    G_CONSTANT: (1, "ConstantGate { num_consts: 2 }"),
    G_PUBLIC_INPUT: (1, "PublicInputGate"),
    G_U32_INTERLEAVE: (2, "U32InterleaveGate { num_ops: 3 }"),
    G_U32_UNINTERLEAVE: (2, "UninterleaveToU32Gate { num_ops: 2 }"),
    G_ARITHMETIC: (3, "ArithmeticGate { num_ops: 20 }"),
    G_U32_ARITHMETIC: (4, "U32ArithmeticGate { num_ops: 3, _phantom: " + _PHANTOM + " }"),
    G_POSEIDON2: (7, "Poseidon2Gate { _phantom: " + _PHANTOM + " }<WIDTH=12>"),
}
# num_constraints of the reference's gates: interleave_u32.rs:229-231 (num_ops * (2 + 32)), uninterleave_to_u32.rs:264-266
# (num_ops * (3 + 64)), arithmetic_u32.rs:167-169 (num_ops * (4 + 32)), poseidon2_gate.rs:425-427 (1 + 4 + 12 * 3 + 22 + 12 * 4 + 12)
GATE_CONSTRAINTS = {G_NOOP: 0, G_CONSTANT: 2, G_PUBLIC_INPUT: 4, G_ARITHMETIC: 20, G_U32_INTERLEAVE: 3 * 34,
                    G_U32_UNINTERLEAVE: 2 * 67, G_U32_ARITHMETIC: 3 * 36, G_POSEIDON2: 123}
MULTI_OPS = {G_U32_INTERLEAVE: 3, G_U32_UNINTERLEAVE: 2, G_U32_ARITHMETIC: 3}    # ops per row (num_ops at 135 wires / 80 routed)
MAX_DEGREE = 9                                       # max_quotient_degree_factor + 1 (upstream selectors.rs)
UNUSED_SELECTOR = 0xFFFFFFFF
NUM_WIRES, NUM_ROUTED, NUM_CONSTANTS, ARITH_OPS = 135, 80, 2, 20   # CircuitConfig::standard_recursion_config()


def root_of_unity(log_n):
    g = 1753635133440165772                    # 7^((p-1)/2^32): two_adic.rs:35
    for _ in range(32 - log_n):
        g = g * g % P
    return g


class MiniCircuit:
    """Targets are ("v", i) virtual or ("w", row, col) wires."""

    def __init__(self):
        self.rows = []                 # [kind, c0, c1]
        self.copies = []
        self.n_virtual = 0
        self.inputs = []
        self.constants = {}            # value -> target
        self.open_arith = {}           # (c0, c1) -> (row, next op)
        self.open_ops = {}             # multi-op gate kind -> (row, next op)   (CircuitBuilder::find_slot)
        self.generators = []           # (kind, c0, c1, aux, deps, outs)

    def virtual(self):
        self.n_virtual += 1
        return ("v", self.n_virtual - 1)

    def input(self):
        t = self.virtual()
        self.inputs.append(t)
        return t

    def constant(self, v):
        if v not in self.constants:
            self.constants[v] = self.virtual()
        return self.constants[v]

    def connect(self, a, b):
        self.copies.append((a, b))

    def arithmetic(self, c0, c1, m0, m1, addend):
        """ArithmeticGate op: out = c0*m0*m1 + c1*addend (one open row per (c0, c1), as CircuitBuilder::find_slot)."""
        self.constant(0)                                   # upstream's special-case analysis touches zero()
        key = (c0, c1)
        if key in self.open_arith:
            row, op = self.open_arith[key]
        else:
            row, op = len(self.rows), 0
            self.rows.append([G_ARITHMETIC, c0, c1])
        if op == ARITH_OPS - 1:
            self.open_arith.pop(key, None)
        else:
            self.open_arith[key] = (row, op + 1)
        for k, t in enumerate((m0, m1, addend)):
            self.connect(t, ("w", row, 4 * op + k))
        return ("w", row, 4 * op + 3)

    def mul(self, a, b):
        return self.arithmetic(1, 0, a, b, a)

    def add(self, a, b):
        return self.arithmetic(1, 1, a, self.constant(1), b)

    # ---- the reference's gates (wire layouts from their `wire_*` accessors)
    def _slot(self, kind):
        if kind in self.open_ops:
            row, op = self.open_ops[kind]
        else:
            row, op = len(self.rows), 0
            self.rows.append([kind, 0, 0])
        if op == MULTI_OPS[kind] - 1:
            self.open_ops.pop(kind, None)
        else:
            self.open_ops[kind] = (row, op + 1)
        return row, op

    def interleave_u32(self, x):
        """gadgets/interleaved_u32.rs:89-111 on a U32InterleaveGate op: wires 2i (x) and 2i + 1 (interleaved) (interleave_u32.rs:45-62)."""
        row, i = self._slot(G_U32_INTERLEAVE)
        self.connect(("w", row, 2 * i), x)
        return ("w", row, 2 * i + 1)

    def uninterleave_to_u32(self, x):
        """UninterleaveToU32Gate op: wires 3i (x), 3i + 1 (evens), 3i + 2 (odds) (uninterleave_to_u32.rs:47-75)."""
        row, i = self._slot(G_U32_UNINTERLEAVE)
        self.connect(("w", row, 3 * i), x)
        return ("w", row, 3 * i + 1), ("w", row, 3 * i + 2)

    def mul_add_u32(self, x, y, z):
        """U32ArithmeticGate op: wires 6i, 6i + 1, 6i + 2 (multiplicands, addend), 6i + 3 / 6i + 4 (low, high) (arithmetic_u32.rs:40-80)."""
        row, i = self._slot(G_U32_ARITHMETIC)
        for k, t in enumerate((x, y, z)):
            self.connect(("w", row, 6 * i + k), t)
        return ("w", row, 6 * i + 3), ("w", row, 6 * i + 4)

    def poseidon2_permute(self, state):
        """Poseidon2Hash::permute_targets (poseidon2.rs:585-609): one Poseidon2Gate row, swap = 0, inputs 0..11, outputs 12..23."""
        row = len(self.rows)
        self.rows.append([G_POSEIDON2, 0, 0])
        self.connect(self.constant(0), ("w", row, 24))
        for i, t in enumerate(state):
            self.connect(t, ("w", row, i))
        return [("w", row, 12 + i) for i in range(12)]

    # ------------------------------------------------------------------ build(): the tables of the blob
    def build(self):
        rows = [list(r) for r in self.rows]
        copies = list(self.copies)
        gens = list(self.generators)
        # PublicInputGate: hash of the (empty) public inputs = 4 zeros; unused wires get RandomValueGenerators
        zero = self.constant(0)
        pi_row = len(rows)
        rows.append([G_PUBLIC_INPUT, 0, 0])
        for i in range(4):
            copies.append((zero, ("w", pi_row, i)))
        for w in range(4, NUM_WIRES):
            gens.append((GEN_RANDOM, 0, 0, w, [], [("w", pi_row, w)]))
        # ConstantGates: constants in increasing order, two per row
        consts = sorted(self.constants.items())
        for k, (val, t) in enumerate(consts):
            if k % NUM_CONSTANTS == 0:
                rows.append([G_CONSTANT, 0, 0])
            row = len(rows) - 1
            rows[row][1 + k % NUM_CONSTANTS] = val
            copies.append((("w", row, k % NUM_CONSTANTS), t))
            gens.append((GEN_CONSTANT, val, 0, 0, [], [("w", row, k % NUM_CONSTANTS)]))
        while len(rows) & (len(rows) - 1):
            rows.append([G_NOOP, 0, 0])
        n = len(rows)
        degree_bits = n.bit_length() - 1
        W = NUM_WIRES

        def tidx(t):
            return n * W + t[1] if t[0] == "v" else t[1] * W + t[2]

        n_targets = n * W + self.n_virtual
        # gate set sorted by (degree, id); one selector group when max_degree + num_gates - 1 <= 9
        kinds = sorted({r[0] for r in rows}, key=lambda k: GATE_ORDER[k])
        gate_index = {k: i for i, k in enumerate(kinds)}
        # selector polynomials (upstream selectors.rs::selector_polynomials): one if everything fits the degree bound, else
        # greedy groups -- gates are taken while (gates in the group) + (degree of the next gate) < max_degree
        degs = [GATE_ORDER[k][0] for k in kinds]
        if degs[-1] + len(kinds) - 1 <= MAX_DEGREE:
            groups = [(0, len(kinds))]
        else:
            groups, start = [], 0
            while start < len(kinds):
                size = 0
                while start + size < len(kinds) and size + degs[start + size] < MAX_DEGREE:
                    size += 1
                assert size > 0
                groups.append((start, start + size))
                start += size
        group_of = [next(g for g, (a, b) in enumerate(groups) if a <= i < b) for i in range(len(kinds))]
        if len(groups) == 1:
            selectors = [[gate_index[r[0]] for r in rows]]
        else:
            selectors = [[gate_index[r[0]] if group_of[gate_index[r[0]]] == g else UNUSED_SELECTOR for r in rows]
                         for g in range(len(groups))]
        const_polys = [[r[1] for r in rows], [r[2] for r in rows]]
        # copy constraints -> partitions -> sigma polynomials
        parent = list(range(n_targets))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for a, b in copies:
            ra, rb = find(tidx(a)), find(tidx(b))
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)           # any representative is valid; this one differs from the product's
        rep = [find(i) for i in range(n_targets)]
        k_is = [pow(7, i, P) for i in range(NUM_ROUTED)]
        w_n = root_of_unity(degree_bits)
        subgroup = [pow(w_n, i, P) for i in range(n)]
        members = {}
        for row in range(n):
            for col in range(NUM_ROUTED):
                members.setdefault(rep[row * W + col], []).append((row, col))
        nxt = {}
        for cyc in members.values():                        # next wire of the partition in (row, col) order, cyclic
            for i, rc in enumerate(cyc):
                nxt[rc] = cyc[(i + 1) % len(cyc)]
        sigmas = [[k_is[nxt[(row, col)][1]] * subgroup[nxt[(row, col)][0]] % P for row in range(n)]
                  for col in range(NUM_ROUTED)]
        # per-row gate generators after the explicit ones; unused ops of an incomplete row are dropped
        used_ops = {row: op for (row, op) in self.open_arith.values()}
        used_ops.update({row: op for (row, op) in self.open_ops.values()})

        def w(row, col):
            return ("w", row, col)

        for row, r in enumerate(rows):
            if r[0] == G_ARITHMETIC:
                for i in range(used_ops.get(row, ARITH_OPS)):
                    gens.append((GEN_ARITHMETIC, r[1], r[2], 0, [("w", row, 4 * i + k) for k in range(3)],
                                 [("w", row, 4 * i + 3)]))
            elif r[0] == G_U32_INTERLEAVE:      # interleave_u32.rs:305-334: x -> 32 bits (wires 6 + 32 i ..), interleaved
                for i in range(used_ops.get(row, 3)):
                    gens.append((GEN_U32_INTERLEAVE, 0, 0, 0, [w(row, 2 * i)],
                                 [w(row, 6 + 32 * i + j) for j in range(32)] + [w(row, 2 * i + 1)]))
            elif r[0] == G_U32_UNINTERLEAVE:    # uninterleave_to_u32.rs:353-390: x -> 64 bits (wires 6 + 64 i ..), evens, odds
                for i in range(used_ops.get(row, 2)):
                    gens.append((GEN_U32_UNINTERLEAVE, 0, 0, 0, [w(row, 3 * i)],
                                 [w(row, 6 + 64 * i + j) for j in range(64)] + [w(row, 3 * i + 1), w(row, 3 * i + 2)]))
            elif r[0] == G_U32_ARITHMETIC:      # arithmetic_u32.rs:389-439: m0, m1, addend -> low, high, inverse, 32 limbs (wires 18 + 32 i ..)
                for i in range(used_ops.get(row, 3)):
                    gens.append((GEN_U32_ARITHMETIC, 0, 0, 0, [w(row, 6 * i + k) for k in range(3)],
                                 [w(row, 6 * i + 3), w(row, 6 * i + 4), w(row, 6 * i + 5)] + [w(row, 18 + 32 * i + j) for j in range(32)]))
            elif r[0] == G_POSEIDON2:           # poseidon2_gate.rs:447-523: 12 inputs, swap -> 4 deltas, 106 S-box inputs, 12 outputs
                gens.append((GEN_POSEIDON2, 0, 0, 0, [w(row, k) for k in range(12)] + [w(row, 24)],
                             [w(row, 25 + k) for k in range(4)] + [w(row, 29 + k) for k in range(106)] + [w(row, 12 + k) for k in range(12)]))
        # FRI schedule: ConstantArityBits(4, 5) under rate_bits 3, cap_height 4
        arity, db = [], degree_bits
        while db > 5 and db + 3 - 4 >= 4:
            arity.append(4)
            db -= 4
        return dict(degree_bits=degree_bits, rows=rows, kinds=kinds, selectors=selectors, groups=groups, group_of=group_of, const_polys=const_polys,
                    sigmas=sigmas, k_is=k_is, rep=rep, gens=gens, pi_row=pi_row, arity=arity,
                    inputs=[tidx(t) for t in self.inputs], tidx=tidx, n_virtual=self.n_virtual)

    # ------------------------------------------------------------------ serialisation
    def to_blob(self):
        b = self.build()
        n = 1 << b["degree_bits"]
        out = bytearray(MAGIC)

        def u64s(vals):
            out.extend(struct.pack(f"<{len(vals)}Q", *vals))

        def u32s(vals):
            out.extend(struct.pack(f"<{len(vals)}I", *vals))
            if len(vals) % 2:
                out.extend(b"\0\0\0\0")                     # u32 arrays are padded to 8 bytes

        kinds = b["kinds"]
        header = [0] * 32
        header[0] = b["degree_bits"]
        header[1], header[2], header[3] = NUM_WIRES, NUM_ROUTED, NUM_CONSTANTS
        header[4], header[5], header[6], header[7] = 2, 8, 3, 4      # challenges, max quotient degree factor, rate_bits, cap_height
        header[8], header[9] = 16, 28                                # proof_of_work_bits, num_query_rounds
        header[10] = len(b["arity"])
        header[11] = len(b["selectors"])                             # selector polynomials
        header[12] = max(GATE_CONSTRAINTS[k] for k in kinds)
        header[13] = -(-NUM_ROUTED // 8) - 1                         # partial products per challenge
        header[14] = len(kinds)
        header[15] = b["pi_row"]
        header[16] = b["n_virtual"]
        header[17] = len(b["inputs"])
        header[18] = len(b["gens"])
        header[19] = len(b["selectors"]) + NUM_CONSTANTS + NUM_ROUTED
        header[20], header[21] = 4, 5                                # FRI ConstantArityBits(4, 5)
        u64s(header)
        for i, k in enumerate(kinds):
            g = b["group_of"][i]
            u64s([k, g, b["groups"][g][0], b["groups"][g][1]])         # kind, selector index, group [start, end)
        u64s(b["arity"])
        u32s([r[0] for r in b["rows"]])
        for p in b["selectors"]:
            u64s(p)
        for p in b["const_polys"]:
            u64s(p)
        for p in b["sigmas"]:
            u64s(p)
        u64s(b["k_is"])
        u32s(b["inputs"])
        u32s(b["rep"])
        for kind, c0, c1, aux, deps, outs in b["gens"]:
            u64s([kind, c0, c1, aux, len(deps), len(outs)])
            u32s([b["tidx"](t) for t in deps] + [b["tidx"](t) for t in outs])
        assert n == len(b["rows"])
        return bytes(out)


def connected_inputs_product():
    """The circuit of p25_circuit_build_gadget(8): inputs a, b, expected; connect(a, b); a*b == expected."""
    c = MiniCircuit()
    a, b, expected = c.input(), c.input(), c.input()
    c.connect(a, b)
    c.connect(c.mul(a, b), expected)
    return c


def sum_of_products(k):
    """A circuit the library has no builder for: inputs x_0..x_{2k-1}, y; sum_i x_{2i}*x_{2i+1} == y (k products and
    k-1 additions: two ArithmeticGate rows with different constants, 2k+1 inputs)."""
    c = MiniCircuit()
    xs = [c.input() for _ in range(2 * k)]
    y = c.input()
    acc = c.mul(xs[0], xs[1])
    for i in range(1, k):
        acc = c.add(acc, c.mul(xs[2 * i], xs[2 * i + 1]))
    c.connect(acc, y)
    return c


def reference_gates():
    """The circuit of p25_circuit_build_gadget(14): the reference's four gates + ArithmeticGate (8 gate types, two selector groups)."""
    c = MiniCircuit()
    x, y, z = c.input(), c.input(), c.input()
    m = c.mul(x, y)
    xi = c.interleave_u32(x)
    yi = c.interleave_u32(y)
    ev, od = c.uninterleave_to_u32(xi)
    lo, hi = c.mul_add_u32(x, y, z)
    zero = c.constant(0)
    out = c.poseidon2_permute([m, xi, yi, ev, od, lo, hi, x, zero, zero, zero, zero])
    for t in (m, xi, yi, ev, od, lo, hi):
        c.connect(t, c.input())
    for i in range(4):
        c.connect(out[i], c.input())
    return c
