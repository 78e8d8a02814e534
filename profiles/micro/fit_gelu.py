"""Fit and fp32 check of fc_gelu (flowcompare_amd/csrc/activations.h): Phi(-|v|) = exp2(g(|v|)) with ONE degree-11 polynomial
g(w) = log2 erfcx(w / sqrt 2) - (w^2 / 2) log2 e - 1 on w in [0, 5.2 sqrt 2], evaluated as the kernel does (fmaf Horner chain in fp32,
w = min(|v|, W), exp2, fma(-|v|, e, max(v, 0))).  Round 3 fitted log2 erfcx(u) in u = |v| / sqrt 2 and added the exponent separately
(20 instructions); round 4 folded the exponent and the 1/2 into the polynomial (16) and the 1 / sqrt 2 into its argument (15)."""
import numpy as np
import numpy.polynomial.chebyshev as C
from numpy.polynomial import polynomial as P
from scipy.special import erf, erfcx

W, DEG = 5.2 * np.sqrt(2.0), 11
xs = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000)
w = (xs + 1) * W / 2
target = np.log(erfcx(w / np.sqrt(2))) * np.log2(np.e) - 0.5 * w * w * np.log2(np.e) - 1.0
p = C.cheb2poly(C.chebfit(xs, target, DEG))
pu, base, lin = np.array([0.0]), np.array([1.0]), np.array([-1.0, 2.0 / W])
for ck in p:
    pu = P.polyadd(pu, ck * base)
    base = P.polymul(base, lin)
co = pu.astype(np.float32)
print("W =", repr(W))
print("coefficients (w^0 ... w^11):", ", ".join(repr(float(c)) for c in co))
f = np.float32
v = np.linspace(-9, 9, 900001).astype(f)
ww = np.minimum(np.abs(v), f(W)).astype(f)
g = np.full_like(ww, co[-1])
for ck in co[-2::-1]:
    g = (g.astype(np.float64) * ww.astype(np.float64) + np.float64(ck)).astype(f)             # fmaf: one rounding
e = np.exp2(g.astype(np.float64)).astype(f)                                                  # Phi(-|v|)
out = (np.maximum(v, f(0)).astype(np.float64) - np.abs(v).astype(np.float64) * e.astype(np.float64)).astype(f)   # fmaf(-|v|, e, max(v, 0))
ref = 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))
err = np.abs(out - ref)
print("max abs", err.max(), "scaled by max(1,|v|)", (err / np.maximum(1, np.abs(v))).max(),
      "max relative on 1e-4 < |v| < 6", (err / np.maximum(np.abs(ref), 1e-30))[(np.abs(v) < 6) & (np.abs(v) > 1e-4)].max())
