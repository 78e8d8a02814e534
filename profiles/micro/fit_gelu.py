"""Fit and fp32 check of fc_gelu (flowcompare_amd/csrc/activations.h): degree-11 polynomial of log2 erfcx(u) on [0, 5.2]."""
import numpy as np
import numpy.polynomial.chebyshev as C
from numpy.polynomial import polynomial as P
from scipy.special import erf, erfcx

U, DEG = 5.2, 11
xs = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000)
u = (xs + 1) * U / 2
p = C.cheb2poly(C.chebfit(xs, np.log(erfcx(u)) * np.log2(np.e), DEG))
pu, base, lin = np.array([0.0]), np.array([1.0]), np.array([-1.0, 2.0 / U])
for ck in p:
    pu = P.polyadd(pu, ck * base)
    base = P.polymul(base, lin)
co = pu.astype(np.float32)
print("coefficients (u^0 ... u^11):", ", ".join(repr(float(c)) for c in co))
f = np.float32
v = np.linspace(-8, 8, 400001).astype(f)
uu = np.minimum(np.abs(v) * f(0.70710678), f(U)).astype(f)
# round 4: the exponent's - u^2 log2 e rides in the u^2 coefficient and the - 1 (the 1/2 of erfc / 2) in the constant, as in fc_gelu
co2 = co.astype(np.float64).copy()
co2[2] = np.float64(f(co[2])) - 1.4426950408889634
co2[0] = np.float64(f(co[0])) - 1.0
g = np.full_like(uu, f(co2[-1]))
for ck in co2[-2::-1]:
    g = (g.astype(np.float64) * uu.astype(np.float64) + np.float64(f(ck))).astype(f)          # fmaf: one rounding
e = np.exp2(g.astype(np.float64)).astype(f)                                                  # erfc(u) / 2
out = (np.maximum(v, f(0)).astype(np.float64) - np.abs(v).astype(np.float64) * e.astype(np.float64)).astype(f)   # fmaf(-|v|, e, max(v, 0))
ref = 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))
err = np.abs(out - ref)
print("max abs", err.max(), "scaled by max(1,|v|)", (err / np.maximum(1, np.abs(v))).max(),
      "max relative on 1e-4 < |v| < 6", (err / np.maximum(np.abs(ref), 1e-30))[(np.abs(v) < 6) & (np.abs(v) > 1e-4)].max())
