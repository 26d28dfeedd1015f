"""Fit P in  gelu(x) = max(x, 0) - |x| 2^P(|x|),  P ~ log2(0.5 erfc(a / sqrt 2)),  by minimising the max-abs error of the
whole expression against fp64 (the form used by det_common.h: gelu1).  Usage: python tools/fit_gelu.py [degree]"""
import sys
import numpy as np
from scipy import optimize, special

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
a = np.concatenate([np.linspace(0.0, 6.5, 20001), np.linspace(6.5, 12.0, 2001)])
phi = 0.5 * special.erfc(a / np.sqrt(2.0))
target = np.log2(phi)
w = a * phi * np.log(2.0)                       # d(gelu)/dP


def err64(c):
    p = np.polyval(c, a)
    return a * np.exp2(p) - a * phi             # error of the correction term (same for both signs of x)


# start: weighted least squares on P, then minimise the max error (smooth-max continuation)
V = np.vander(a, deg + 1)
c0 = np.linalg.lstsq(V * (w + 1e-12)[:, None], target * (w + 1e-12), rcond=None)[0]
c = c0
for beta in (1e5, 1e6, 1e7, 1e8):
    f = lambda c: np.log(np.sum(np.exp(beta * np.abs(err64(c)) - beta * np.abs(err64(c)).max()))) / beta + np.abs(err64(c)).max()
    c = optimize.minimize(f, c, method="Nelder-Mead", options=dict(xatol=1e-14, fatol=1e-16, maxiter=40000, maxfev=40000)).x
print("degree", deg, "max |err| fp64 eval:", np.abs(err64(c)).max())

# fp32 evaluation as the kernel does it (Horner with fma ~ float32 ops, v_exp_f32 ~ exp2 in float32)
c32 = c.astype(np.float32)
x = np.linspace(-12, 12, 400001).astype(np.float32)
ax = np.abs(x)
p = np.full_like(ax, c32[0])
for k in c32[1:]:
    p = (p.astype(np.float64) * ax + k).astype(np.float32)          # fma: one rounding
e = np.exp2(p.astype(np.float64)).astype(np.float32)
y = (np.maximum(x, 0).astype(np.float64) - ax.astype(np.float64) * e).astype(np.float32)
ref = x.astype(np.float64) * 0.5 * special.erfc(-x.astype(np.float64) / np.sqrt(2.0))
print("max |err| fp32 eval over [-12, 12]:", np.abs(y - ref).max())
print("coefficients (highest degree first):", ", ".join(f"{v:.9e}f" for v in c32))
