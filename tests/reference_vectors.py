"""Test DATA the reference's own unit tests hold for the hot path's gates and gadgets, restated as values (not code):

* `test_interleave_u32`        /root/reference/src/common/u32/gadgets/interleaved_u32.rs:354-382
* `test_uninterleave_to_u32`   .../interleaved_u32.rs:388-417
* `get_wires` / `test_gate_constraint` / `test_canonicity`   /root/reference/src/common/u32/gates/arithmetic_u32.rs:501-627
* `wire_indices`               /root/reference/src/common/poseidon2/poseidon2_gate.rs:553-565
"""
import numpy as np

P = 0xFFFFFFFF00000001

# interleaved_u32.rs:365-369 (binary literals written out)
INTERLEAVE_X = 0b1111_1111_1111_1111_1111_1111_1111_1100
INTERLEAVE_EXPECTED = 0b0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0000
# interleaved_u32.rs:398-401
UNINTERLEAVE_X = 0b1111_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101_0101
UNINTERLEAVE_EVENS_EXPECTED = 0b1100_0000_0000_0000_0000_0000_0000_0000
UNINTERLEAVE_ODDS_EXPECTED = 0b1111_1111_1111_1111_1111_1111_1111_1111
assert INTERLEAVE_EXPECTED == 0x5555555555555550 and UNINTERLEAVE_X == 0xF555555555555555

# arithmetic_u32.rs:594-600: "A non-canonical addend will produce a non-canonical output using get_wires."
CANONICITY_ADDEND = 0xFFFFFFFF00000001

# poseidon2_gate.rs:553-565
POSEIDON2_WIRE_PINS = {"wire_input(0)": 0, "wire_input(11)": 11, "wire_output(0)": 12, "wire_output(11)": 23,
                       "WIRE_SWAP": 24, "wire_delta(0)": 25, "wire_delta(3)": 28}

U32_ARITH_OPS, U32_ARITH_LIMB_BITS, U32_ARITH_LIMBS = 3, 2, 32     # arithmetic_u32.rs:52-60 (num_ops 3 at 135 wires), :95-100
U32_ARITH_WIRES = 6 * U32_ARITH_OPS + U32_ARITH_LIMBS * U32_ARITH_OPS


def u32_arithmetic_get_wires(multiplicands_0, multiplicands_1, addends):
    """The wire VALUES the reference's test helper `get_wires` (arithmetic_u32.rs:501-552) assigns for NUM_OPS = 3:
    per op (m0, m1, addend [from_noncanonical_u64], output_low, output_high, inverse of (u32::MAX - output_high) or 0),
    then every op's 32 two-bit limbs of the 64-bit output, little-endian.  u64 arithmetic as in the Rust helper."""
    v0, v1 = [], []
    for m0, m1, a in zip(multiplicands_0, multiplicands_1, addends):
        output = m0 * m1 + a
        assert output < 1 << 64                       # the helper's u64 arithmetic does not wrap on the tests' inputs
        lo, hi = output & 0xFFFFFFFF, output >> 32
        diff = 0xFFFFFFFF - hi
        inv = 0 if diff == 0 else pow(diff, P - 2, P)
        v0 += [m0 % P, m1 % P, a % P, lo, hi, inv]
        for _ in range(U32_ARITH_LIMBS):
            v1.append(output % 4)
            output //= 4
    return np.array(v0 + v1, dtype=np.uint64)


def u32_arithmetic_cases(seed=20241004, n_random=6):
    """(name, wires[114], satisfied?): test_gate_constraint's distribution (three random u32 triples; the reference
    draws from OsRng, here seeded) plus edge triples, and test_canonicity's literal input."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_random):
        m0, m1, ad = ([int(v) for v in rng.integers(0, 1 << 32, size=3)] for _ in range(3))
        out.append((f"random{t}", u32_arithmetic_get_wires(m0, m1, ad), True))
    M = 0xFFFFFFFF
    out.append(("edges", u32_arithmetic_get_wires([0, M, M], [0, M, 1], [0, M, 0]), True))     # M*M + M: output_high = u32::MAX
    out.append(("canonicity", u32_arithmetic_get_wires([0] * 3, [0] * 3, [CANONICITY_ADDEND] * 3), False))
    return out
