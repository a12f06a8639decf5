"""Derive the float32 polynomial coefficients of the engine's canonical elementary functions
(sin on [-pi/2,pi/2], exp on [-ln2/2,ln2/2], tanh on [-0.55,0.55]).  Weighted least squares on
Chebyshev nodes in float64 (near-minimax), then rounded to float32 and printed as C hex-floats.
The printed constants are pasted into neuralcodecs_amd/csrc/nc_math.h and oracle/c/ref_math.h."""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    return 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * (2 * k + 1) / (2 * n))

def fit(fun, a, b, deg, n=4000):
    x = cheb_nodes(a, b, n)
    V = np.vander(x, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V, fun(x), rcond=None)
    return c

def show(name, c):
    c32 = c.astype(np.float32)
    print(name, ", ".join(float(v).hex() + "f" for v in c32))
    return c32

# sin(r) = r + r^3 * P(r^2),  P(u) = (sin(sqrt(u))/sqrt(u) - 1)/u
def fs(u):
    r = np.sqrt(u)
    return (np.sin(r) / r - 1.0) / u
cs = show("SIN", fit(fs, 1e-12, (np.pi / 2) ** 2 * 1.02, 4))
# exp(r) = 1 + r + r^2 * Q(r)
def fe(r):
    return np.where(np.abs(r) < 1e-6, 0.5 + r / 6, (np.exp(r) - 1 - r) / (r * r))
ce = show("EXP", fit(fe, -0.36, 0.36, 4))
# tanh(x) = x + x^3 * T(x^2)
def ft(u):
    x = np.sqrt(u)
    return (np.tanh(x) / x - 1.0) / u
ct = show("TANH", fit(ft, 1e-12, 0.56 ** 2, 4))

# accuracy check in float32 arithmetic emulation
def f32(x): return np.float32(x)
def sin32(x):
    x = x.astype(np.float32)
    n = np.rint(x * f32(0.318309886))
    r = (x.astype(np.float64) - n * 3.140625 - n * 9.67502593994140625e-4 - n * 1.509957990978376432e-7).astype(np.float32)
    u = r * r
    p = cs[4]
    for k in (3, 2, 1, 0):
        p = (p.astype(np.float64) * u + cs[k]).astype(np.float32)
    s = ((r * u).astype(np.float64) * p + r).astype(np.float32)
    return np.where(n.astype(np.int64) & 1, -s, s)
xs = np.linspace(-40, 40, 2000001)
err = np.abs(sin32(xs).astype(np.float64) - np.sin(xs.astype(np.float32).astype(np.float64)))
print("sin max abs err", err.max(), "in ulp(1)=", err.max() / 2 ** -24)
