"""Coefficients of gelu_erf's error function (csrc/encoder_kernels.hpp): q(t) = log(erfc(t)) on [0, 4] as a degree-9
polynomial, fitted so that max |erfc(t) dq(t)| -- the ABSOLUTE error of erf(t) = 1 - exp(q(t)) -- is smallest (iteratively
re-weighted least squares on Chebyshev nodes).  GELU needs erf to an absolute accuracy near one fp32 ulp of 1, not a
relative one; one polynomial + one v_exp_f32 replaces the two-branch library erff (~45 instructions on a divergent wave).
    python tools/fit_gelu_erf.py          prints the coefficients (lowest order first) and the errors"""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf, erfc

T, DEG = 4.0, 9


def fit():
    t = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * T / 2 + T / 2
    q = np.log(erfc(t))
    xs = (t - T / 2) / (T / 2)
    w = erfc(t).copy()
    for _ in range(60):
        c = C.chebfit(xs, q, DEG, w=w)
        r = np.abs((C.chebval(xs, c) - q) * erfc(t))
        w = w * (1 + 2 * r / r.max())
        w /= w.max()
    p = np.poly1d(C.cheb2poly(c)[::-1])(np.poly1d([2 / T, -1]))       # Chebyshev series in 2t/T - 1 -> monomials in t
    return p.coeffs[::-1]


def main():
    co = fit()
    tt = np.linspace(0, T, 400001)
    q64 = np.polyval(co[::-1], tt)
    print("max |erf - (1 - exp(q))|, float64 evaluation: %.3g" % np.abs((1 - np.exp(q64)) - erf(tt)).max())
    t32 = tt.astype(np.float32)
    acc = np.full_like(t32, np.float32(co[-1]))
    for a in co[-2::-1]:                                              # Horner, every operation rounded to fp32
        acc = ((acc.astype(np.float64) * t32 + np.float64(np.float32(a)))).astype(np.float32)
    e32 = (np.float32(1) - np.exp2((acc * np.float32(1.4426950408889634)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    print("fp32 Horner (fma) + exp2: %.3g" % np.abs(e32.astype(np.float64) - erf(tt)).max())
    print("erfc(%g) = %.3g (|t| >= %g is clamped)" % (T, erfc(T), T))
    print("{" + ", ".join("%.9gf" % np.float32(a) for a in co) + "}")


if __name__ == "__main__":
    main()
