#!/usr/bin/env python3
"""Near-minimax polynomial coefficients for the two inverse-trig helpers of the SHOT kernel
(shot_fpfh_amd/csrc/shot.hip: sf_atan_small, sf_acos).

    atan(t) = t * P(t^2)           t in [0, tan(pi/8) (1 + 1e-3)]
    asin(r) = r + r * s * R(s)     s = r^2 in [0, 0.25]

Chebyshev interpolation in 60-digit arithmetic (mpmath), coefficients rounded to double, then the
error of the double-precision Horner form is measured on a dense grid.  Prints C initialisers.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    n = deg + 1
    nodes = [mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    xs = [(a + b) / 2 + (b - a) / 2 * t for t in nodes]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [float(c[j]) for j in range(n)]


def horner(c, s):
    acc = np.full_like(s, c[-1])
    for k in reversed(c[:-1]):
        acc = acc * s + k  # numpy: not fused, slightly pessimistic
    return acc


def atan_q(s):
    return mp.mpf(1) if s == 0 else mp.atan(mp.sqrt(s)) / mp.sqrt(s)


def asin_r(s):
    if s == 0:
        return mp.mpf(1) / 6
    r = mp.sqrt(s)
    return (mp.asin(r) - r) / (r * s)


def report(name, c, f_true, grid, build):
    approx = build(grid)
    true = np.array([float(f_true(mp.mpf(float(x)))) for x in grid])
    err = np.abs(approx - true).max()
    print(f"// {name}: degree {len(c) - 1}, max abs error {err:.3e} on {len(grid)} points")
    print("{" + ", ".join(f"{v!r}" for v in c) + "}")


if __name__ == "__main__":
    tmax = float(mp.tan(mp.pi / 8)) * 1.001
    for deg in (10, 11, 12):
        c = cheb_fit(atan_q, mp.mpf(0), mp.mpf(tmax) ** 2, deg)
        t = np.linspace(0.0, tmax, 20001)
        report("atan(t) = t*P(t^2)", c, mp.atan, t, lambda t: t * horner(c, t * t))
    for deg in (11, 12, 13, 14):
        c = cheb_fit(asin_r, mp.mpf(0), mp.mpf("0.25"), deg)
        r = np.linspace(0.0, 0.5, 20001)
        report("asin(r) = r + r*s*R(s)", c, mp.asin, r, lambda r: r + r * (r * r) * horner(c, r * r))
