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
GEN_CONSTANT, GEN_RANDOM, GEN_ARITHMETIC = 0, 1, 2
# (degree, upstream Gate::id()) -- orders the gate set
GATE_ORDER = {
    G_NOOP: (0, "NoopGate"),
    G_CONSTANT: (1, "ConstantGate { num_consts: 2 }"),
    G_PUBLIC_INPUT: (1, "PublicInputGate"),
    G_ARITHMETIC: (3, "ArithmeticGate { num_ops: 20 }"),
}
GATE_CONSTRAINTS = {G_NOOP: 0, G_CONSTANT: 2, G_PUBLIC_INPUT: 4, G_ARITHMETIC: 20}
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
        assert GATE_ORDER[kinds[-1]][0] + len(kinds) - 1 <= 9
        gate_index = {k: i for i, k in enumerate(kinds)}
        selector = [gate_index[r[0]] for r in rows]
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
        for row, r in enumerate(rows):
            if r[0] == G_ARITHMETIC:
                for i in range(used_ops.get(row, ARITH_OPS)):
                    gens.append((GEN_ARITHMETIC, r[1], r[2], 0, [("w", row, 4 * i + k) for k in range(3)],
                                 [("w", row, 4 * i + 3)]))
        # FRI schedule: ConstantArityBits(4, 5) under rate_bits 3, cap_height 4
        arity, db = [], degree_bits
        while db > 5 and db + 3 - 4 >= 4:
            arity.append(4)
            db -= 4
        return dict(degree_bits=degree_bits, rows=rows, kinds=kinds, selector=selector, const_polys=const_polys,
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
        header[11] = 1                                               # selector polynomials
        header[12] = max(GATE_CONSTRAINTS[k] for k in kinds)
        header[13] = -(-NUM_ROUTED // 8) - 1                         # partial products per challenge
        header[14] = len(kinds)
        header[15] = b["pi_row"]
        header[16] = b["n_virtual"]
        header[17] = len(b["inputs"])
        header[18] = len(b["gens"])
        header[19] = 1 + NUM_CONSTANTS + NUM_ROUTED
        header[20], header[21] = 4, 5                                # FRI ConstantArityBits(4, 5)
        u64s(header)
        for i, k in enumerate(kinds):
            u64s([k, 0, 0, len(kinds)])                              # kind, selector index, group [start, end)
        u64s(b["arity"])
        u32s([r[0] for r in b["rows"]])
        u64s(b["selector"])
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
