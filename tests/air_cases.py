"""User AIRs for the 8f-2 tests (data form of the reference's `Air` trait, src/p3/air.rs:10-18) with
trace generators.  Constraint degree (selector included) stays <= 2: one quotient chunk."""
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
