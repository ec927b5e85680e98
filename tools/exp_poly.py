"""Coefficients of exp_bounded (bayes.js_amd/csrc/amwg_math.h ExpTaylorLiterals): the degree-11 polynomial that interpolates exp at the 12 Chebyshev nodes of
[-a, a], a = ln2 / 2 (1 + 1e-7), in 60-digit arithmetic, converted to the monomial basis and rounded to doubles (c0 and c1 round to 1 exactly).  Prints the doubles and
the polynomial's distance from exp in EXACT arithmetic on a grid -- 1.7e-17 relative: what is left of exp_bounded's error is its eleven fused steps' roundings.
python tools/exp_poly.py"""
from decimal import Decimal as D, getcontext
getcontext().prec = 60
import math
def dcos(x):  # Taylor, x Decimal
    x = D(x); s = D(1); t = D(1); k = 0
    while abs(t) > D(10) ** -58:
        k += 2; t = -t * x * x / (k * (k - 1)); s += t
    return s
PI = D("3.14159265358979323846264338327950288419716939937510582097494")
a = D("0.34657359027997265")  # ln2/2 (slightly above)
a = a * D("1.0000001")
N = 12  # degree 11 interpolant: 12 nodes
nodes = [dcos((2 * j + 1) * PI / (2 * N)) for j in range(N)]
f = [(a * t).exp() for t in nodes]
# Chebyshev coefficients
def T(k, t):
    if k == 0: return D(1)
    if k == 1: return t
    a0, a1 = D(1), t
    for _ in range(k - 1): a0, a1 = a1, 2 * t * a1 - a0
    return a1
c = []
for k in range(N):
    s = sum(f[j] * T(k, nodes[j]) for j in range(N)) * 2 / N
    c.append(s)
c[0] /= 2
# convert to monomial in t: sum c_k T_k(t)
polys = [[D(1)], [D(0), D(1)]]
for k in range(2, N):
    p = [D(0)] + [2 * v for v in polys[k - 1]]
    q = polys[k - 2] + [D(0)] * (len(p) - len(polys[k - 2]))
    polys.append([p[i] - q[i] for i in range(len(p))])
mono = [D(0)] * N
for k in range(N):
    for i, v in enumerate(polys[k]): mono[i] += c[k] * v
# in r = a t: coefficient of r^i = mono[i] / a^i
coef = [mono[i] / a ** i for i in range(N)]
for i, v in enumerate(coef):
    print(i, float(v).hex(), float(v), "x %d! = %.16g" % (i, float(v * math.factorial(i))))
cd = [D(float(v)) for v in coef]      # the coefficients as the doubles they become
worst = D(0)
for q in range(-4000, 4001):
    r = D("0.34657359027997265") * D(q) / 4000
    p = D(0)
    for v in reversed(cd):
        p = p * r + v
    worst = max(worst, abs(p / r.exp() - 1))
print("max relative distance from exp, double coefficients, exact arithmetic:", float(worst))
print("constexpr double " + ", ".join("c%d = %s" % (i, float(coef[i]).hex()) for i in range(2, 12)) + ";")
