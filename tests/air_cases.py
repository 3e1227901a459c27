"""User AIRs for the 8f-2 tests (data form of the reference's `Air` trait, src/p3/air.rs:10-18) with
trace generators.  Constraint degree (selector included) <= 2: one quotient chunk, what the reference's proof model holds
(serde/proof.rs:41-48); the `cubic*` AIRs at the end have degree 3: two chunks (round 5)."""
import numpy as np

P = 0xFFFFFFFF00000001


def fib_trace(log_n):
    n = 1 << log_n
    t = np.zeros((n, 3), dtype=np.uint64)
    a, b = 1, 1
    for i in range(n):
        c = (a + b) % P
        t[i] = (a, b, c)
        a, b = b, c
    return t


def tribonacci(p25):
    """width 4: d = a + b + c on every row; (a, b, c) <- (b, c, d); first row (1, 1, 2); last row pins nothing."""
    air = p25.Air(4)
    a, b, c, d = (air.local(i) for i in range(4))
    na, nb, nc = air.next(0), air.next(1), air.next(2)
    air.assert_zero(air.sub(air.add(air.add(a, b), c), d))
    one, two = air.const(1), air.const(2)
    air.when_first_row(air.sub(a, one))
    air.when_first_row(air.sub(b, one))
    air.when_first_row(air.sub(c, two))
    air.when_transition(air.sub(na, b))
    air.when_transition(air.sub(nb, c))
    air.when_transition(air.sub(nc, d))
    return air


def tribonacci_trace(log_n):
    n = 1 << log_n
    t = np.zeros((n, 4), dtype=np.uint64)
    a, b, c = 1, 1, 2
    for i in range(n):
        d = (a + b + c) % P
        t[i] = (a, b, c, d)
        a, b, c = b, c, d
    return t


def squares(p25):
    """width 3 with a quadratic always-constraint and a last-row constraint:
    s = x * x on every row; x <- x + 3; first row x = 5; last row: y = x (y free elsewhere)."""
    air = p25.Air(3)
    x, s, y = air.local(0), air.local(1), air.local(2)
    nx = air.next(0)
    air.assert_zero(air.sub(air.mul(x, x), s))
    air.when_first_row(air.sub(x, air.const(5)))
    air.when_transition(air.sub(nx, air.add(x, air.const(3))))
    air.when_last_row(air.sub(y, x))
    return air


def squares_trace(log_n, seed=1):
    n = 1 << log_n
    rng = np.random.default_rng(seed)
    t = np.zeros((n, 3), dtype=np.uint64)
    x = 5
    for i in range(n):
        t[i] = (x, (x * x) % P, int(rng.integers(0, P, dtype=np.uint64)))
        x = (x + 3) % P
    t[n - 1, 2] = t[n - 1, 0]
    return t


def random_recurrence(p25, seed, width):
    """A seeded FAMILY of AIRs of any width (docs/DEVELOPER-GUIDE.md's recipe, generated): column j of the next row is
    b_j * local[r_j] + c_j with random r_j, b_j, c_j (a transition constraint per column), the first row is pinned to
    random constants.  Returns (air, coefficients); random_recurrence_trace builds the trace from the coefficients."""
    rng = np.random.default_rng(seed)
    air = p25.Air(width)
    coef = []
    for j in range(width):
        rj = int(rng.integers(0, width))
        bj, cj, first = (int(v) for v in rng.integers(1, P, size=3, dtype=np.uint64))
        coef.append((rj, bj, cj, first))
        air.when_first_row(air.sub(air.local(j), air.const(first)))
        air.when_transition(air.sub(air.next(j), air.add(air.mul(air.const(bj), air.local(rj)), air.const(cj))))
    return air, coef


def random_recurrence_trace(coef, log_n):
    n, width = 1 << log_n, len(coef)
    t = np.zeros((n, width), dtype=np.uint64)
    row = [c[3] for c in coef]
    for i in range(n):
        t[i] = row
        row = [(c[1] * row[c[0]] + c[2]) % P for c in coef]
    return t


def quadratic_pair(p25, seed):
    """width 4, seeded: (x, y) evolve linearly with random coefficients, s = x * y and u = (x + y) * (x + k) hold on EVERY
    row (degree-2 always-constraints).  Exercises two independent quadratic constraints with random constants through
    the verifier circuit's constraint folding."""
    rng = np.random.default_rng(seed)
    a, b, c, d, e, f, k, x0, y0 = (int(v) for v in rng.integers(1, P, size=9, dtype=np.uint64))
    air = p25.Air(4)
    x, y, s, u = (air.local(i) for i in range(4))
    air.assert_zero(air.sub(air.mul(x, y), s))
    air.assert_zero(air.sub(air.mul(air.add(x, y), air.add(x, air.const(k))), u))
    air.when_first_row(air.sub(x, air.const(x0)))
    air.when_first_row(air.sub(y, air.const(y0)))
    air.when_transition(air.sub(air.next(0), air.add(air.add(air.mul(air.const(a), x), air.mul(air.const(b), y)), air.const(c))))
    air.when_transition(air.sub(air.next(1), air.add(air.add(air.mul(air.const(d), x), air.mul(air.const(e), y)), air.const(f))))
    return air, (a, b, c, d, e, f, k, x0, y0)


def quadratic_pair_trace(par, log_n):
    a, b, c, d, e, f, k, x, y = par
    n = 1 << log_n
    t = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        t[i] = (x, y, x * y % P, (x + y) * (x + k) % P)
        x, y = (a * x + b * y + c) % P, (d * x + e * y + f) % P
    return t


def cubic(p25):
    """width 2, degree 3 in an ALWAYS constraint: y = x^3 on every row; x <- y + 1; first row x = 2.  Two quotient chunks."""
    air = p25.Air(2)
    x, y = air.local(0), air.local(1)
    air.assert_zero(air.sub(air.mul(air.mul(x, x), x), y))
    air.when_first_row(air.sub(x, air.const(2)))
    air.when_transition(air.sub(air.next(0), air.add(y, air.const(1))))
    return air


def cubic_trace(log_n):
    n = 1 << log_n
    t = np.zeros((n, 2), dtype=np.uint64)
    x = 2
    for i in range(n):
        y = pow(x, 3, P)
        t[i] = (x, y)
        x = (y + 1) % P
    return t


def cubic_transition(p25):
    """width 3, degree 3 through the SELECTOR: the transition constraint next x = x * y + 5 is quadratic, times
    is_transition; y <- y + x (linear), z = x * y on every row (degree 2), last row: z pinned to x * y through a second
    route (z - x y is already zero: the constraint z - x*y under when_last_row has degree 3 as well)."""
    air = p25.Air(3)
    x, y, z = air.local(0), air.local(1), air.local(2)
    air.assert_zero(air.sub(air.mul(x, y), z))
    air.when_first_row(air.sub(x, air.const(3)))
    air.when_first_row(air.sub(y, air.const(7)))
    air.when_transition(air.sub(air.next(0), air.add(air.mul(x, y), air.const(5))))
    air.when_transition(air.sub(air.next(1), air.add(y, x)))
    air.when_last_row(air.sub(air.mul(x, y), z))
    return air


def cubic_transition_trace(log_n):
    n = 1 << log_n
    t = np.zeros((n, 3), dtype=np.uint64)
    x, y = 3, 7
    for i in range(n):
        t[i] = (x, y, x * y % P)
        x, y = (x * y + 5) % P, (y + x) % P
    return t


def quartic_map(p25, seed):
    """Seeded family, constraint degree 4 in an ALWAYS constraint (FOUR quotient chunks, needs log_blowup >= 2):
    width 2, y = x^4 + a x + b on every row; next x = y + c x (transition); first row x = x0.  Returns (air, (a, b, c, x0))."""
    rng = np.random.default_rng(seed)
    a, b, c, x0 = (int(v) for v in rng.integers(1, P, size=4, dtype=np.uint64))
    air = p25.Air(2)
    x, y = air.local(0), air.local(1)
    x2 = air.mul(x, x)
    x4 = air.mul(x2, x2)
    air.assert_zero(air.sub(air.add(air.add(x4, air.mul(air.const(a), x)), air.const(b)), y))
    air.when_first_row(air.sub(x, air.const(x0)))
    air.when_transition(air.sub(air.next(0), air.add(y, air.mul(air.const(c), x))))
    return air, (a, b, c, x0)


def quartic_map_trace(par, log_n):
    a, b, c, x = par
    n = 1 << log_n
    t = np.zeros((n, 2), dtype=np.uint64)
    for i in range(n):
        y = (pow(x, 4, P) + a * x + b) % P
        t[i] = (x, y)
        x = (y + c * x) % P
    return t


def quintic_selector(p25, seed):
    """Seeded family, constraint degree 5 THROUGH THE SELECTOR (four quotient chunks): width 3, the transition constraint
    next x = x^2 y^2 + k is quartic, times is_transition; next y = y + d x; z = x y on every row (degree 2); the last row pins
    z - x y once more under when_last_row (degree 3).  Returns (air, (k, d, x0, y0))."""
    rng = np.random.default_rng(seed)
    k, d, x0, y0 = (int(v) for v in rng.integers(1, P, size=4, dtype=np.uint64))
    air = p25.Air(3)
    x, y, z = air.local(0), air.local(1), air.local(2)
    air.assert_zero(air.sub(air.mul(x, y), z))
    air.when_first_row(air.sub(x, air.const(x0)))
    air.when_first_row(air.sub(y, air.const(y0)))
    xy = air.mul(x, y)
    air.when_transition(air.sub(air.next(0), air.add(air.mul(xy, xy), air.const(k))))
    air.when_transition(air.sub(air.next(1), air.add(y, air.mul(air.const(d), x))))
    air.when_last_row(air.sub(air.mul(x, y), z))
    return air, (k, d, x0, y0)


def quintic_selector_trace(par, log_n):
    k, d, x, y = par
    n = 1 << log_n
    t = np.zeros((n, 3), dtype=np.uint64)
    for i in range(n):
        t[i] = (x, y, x * y % P)
        x, y = (pow(x * y % P, 2, P) + k) % P, (y + d * x) % P
    return t
