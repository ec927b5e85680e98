"""-m gpu: the kernel's arithmetic building blocks, evaluated ON THE DEVICE, against V8 / IEEE."""
import os

import numpy as np
import pytest

import amwg_ctypes as A
import golden_io
import oracle_lib
import synth

pytestmark = pytest.mark.gpu


def test_device_exp_log_bit_exact_vs_v8():
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math_pairs.bin"), dtype="<f8").reshape(-1, 3)
    e = A.device_eval(0, a[:, 0])
    l = A.device_eval(1, np.abs(a[:, 0]))
    assert e.tobytes() == np.ascontiguousarray(a[:, 1]).tobytes()
    assert l.tobytes() == np.ascontiguousarray(a[:, 2]).tobytes()


def test_device_full_fdlibm_flow_vs_v8():
    """exp_v8_full / log_v8_full (every branch of fdlibm, the yardstick of the fused test below) on the device against V8."""
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert A.device_eval(28, a[:, 0]).tobytes() == np.ascontiguousarray(a[:, 1]).tobytes()
    assert A.device_eval(29, np.abs(a[:, 0])).tobytes() == np.ascontiguousarray(a[:, 2]).tobytes()


def _explog_arguments():
    """as tests/host/explog_fuzz.cpp: the range a log link produces, every high word next to the thresholds of exp's argument reduction,
    and arguments whose exp() lands next to the significand thresholds of log"""
    rng = np.random.default_rng(4242)
    parts = [rng.uniform(-30, 30, 1_000_000), rng.uniform(-745.5, 710, 500_000),
             np.ldexp(rng.uniform(0.5, 1, 300_000), rng.integers(-60, 11, 300_000)) * rng.choice([-1.0, 1.0], 300_000)]
    for h in (0x3fd62e42, 0x3ff0a2b2, 0x3e300000, 0x40862000, 0x40862e42, 0x3ff00000, 0x40874910, 0x3fe62e42, 0x3ff62e42):
        for d in range(-3, 4):
            lo = rng.integers(0, 2 ** 32, 4000, dtype=np.uint64)
            lo[:8] = [0, 0xffffffff, 0xfefa39ef, 0xfefa39ee, 0xfefa39f0, 0x3f3bab73, 0x3f3bab72, 0x3f3bab74]
            u = (np.uint64(h + d) << np.uint64(32)) | lo
            parts += [u.view(np.float64), -u.view(np.float64)]
    for h in (0x3ff6a09c, 0x3fe6a09c, 0x3ff6147a, 0x3fe6b851, 0x3ff6b851, 0x3fe6147a, 0x3ff00000, 0x3feffffe, 0x3ff6a09e, 0x3fe6a09e):
        for d in range(-4, 5):
            m = ((np.uint64(h + d) << np.uint64(32)) | rng.integers(0, 2 ** 32, 6000, dtype=np.uint64)).view(np.float64)
            x = rng.integers(-30, 31, 6000) * 0.6931471805599453 + np.log(m)
            parts += [x, np.nextafter(x, 1e300), np.nextafter(x, -1e300)]
    parts.append(np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 709.782712893384, 709.7827128933841, -745.1332191019411,
                           -745.1332191019412, -708.0, 708.0, 1e-300, -1e-300, 5e-324, 0.34657359027997264, -0.34657359027997264,
                           1.0397207708399179, -1.0397207708399179]))
    return np.concatenate(parts)


def test_device_fused_exp_log_equals_full_flow():
    """exp_log_v8 (the Poisson pass: lambda = exp(eta) and log(lambda) sharing the argument reduction) and the one-formula k of exp_v8,
    on the device, against the full fdlibm control flow on the device -- incl. the slivers where the shortcuts do not apply."""
    x = _explog_arguments()
    want_e, want_l = A.device_eval(28, x), A.device_eval(30, x)
    same = lambda p, q: np.array_equal(p.view(np.uint64)[~np.isnan(q)], q.view(np.uint64)[~np.isnan(q)]) and np.array_equal(np.isnan(p), np.isnan(q))
    assert same(A.device_eval(0, x), want_e)
    assert same(A.device_eval(27, x), want_e)
    assert same(A.device_eval(26, x), want_l)
    assert same(A.device_eval(31, x), want_l)


def test_device_sqrt_correctly_rounded():
    k = np.arange(1, 200001, dtype=np.float64)       # batch_count values, mcmc.js:542
    assert A.device_eval(2, k).tobytes() == np.sqrt(k).tobytes()
    r = np.random.default_rng(1).uniform(0, 1e6, 200000)
    assert A.device_eval(2, r).tobytes() == np.sqrt(r).tobytes()


def test_division_by_invariant_equals_ieee():
    rng = np.random.default_rng(2)
    n = 2_000_000
    a = np.concatenate([rng.uniform(0, 1e4, n), np.exp(rng.uniform(-300, 300, n)), [0.0, 1.0, 3.0, 2.0 ** -450, 2.0 ** 400]])
    b = np.concatenate([rng.uniform(1e-3, 1e3, n), np.exp(rng.uniform(-130, 130, n)), [7.0, 3.0, 3.0, 2.0 ** 190, 2.0 ** -190]])
    # adversarial divisors: significand all ones / just above a power of two
    b[:1000] = np.nextafter(2.0 ** rng.integers(-5, 5, 1000).astype(np.float64), 0)
    b[1000:2000] = np.nextafter(2.0 ** rng.integers(-5, 5, 1000).astype(np.float64), 1e300)
    fast = A.device_eval(4, a, b)
    ieee = A.device_eval(5, a, b)
    assert ieee.tobytes() == (a / b).tobytes()        # device '/' is IEEE
    assert fast.tobytes() == ieee.tobytes()


def test_unscaled_quotient_of_exp_and_log_equals_ieee():
    """quot_plain (amwg_math.h): the division sequence without v_div_scale / v_div_fixup, used for (r*c)/(2-c) of exp and
    f/(2+f) of log, against `/` on the device -- denominators in (1.6, 2.45), numerators from 2^-160 up to 0.5 in magnitude,
    zero, and significands of all ones / one bit."""
    rng = np.random.default_rng(99)
    n = 2_000_000
    den = rng.uniform(1.6, 2.45, n)
    num = np.ldexp(rng.uniform(0.5, 1.0, n), rng.integers(-160, 0, n)) * rng.choice([-1.0, 1.0], n)
    num[:1000] = 0.0
    num[1000:2000] = np.ldexp(1.0 - 2.0 ** -53, rng.integers(-60, 0, 1000))       # all-ones significands
    den[2000:3000] = np.nextafter(2.0, 0) * np.ones(1000)
    den[3000:4000] = 2.0
    num[4000:5000] = np.ldexp(1.0, rng.integers(-60, 0, 1000))
    # the operands log really produces: f = m - 1 over the whole significand range
    m = np.ldexp(rng.uniform(1.0, 2.0, n // 2), 0)
    m = np.where(m > np.sqrt(2.0), m / 2, m)
    f = m - 1.0
    num = np.concatenate([num, f]); den = np.concatenate([den, 2.0 + f])
    fast = A.device_eval(19, num, den)
    ieee = A.device_eval(5, num, den)
    assert fast.tobytes() == ieee.tobytes()
    assert ieee.tobytes() == (num / den).tobytes()


def test_device_ld_and_helpers_match_oracle():
    L = oracle_lib.lib()
    rng = np.random.default_rng(3)
    x = rng.uniform(0.1, 200, 5000)
    lg = A.device_eval(3, x)
    assert lg.tobytes() == np.array([L.orc_lgamma(v) for v in x]).tobytes()
    m, sd = rng.normal(0, 5, 5000), rng.uniform(0.1, 9, 5000)
    got = A.device_eval(6, x, m, sd)
    assert got.tobytes() == np.array([L.orc_ld_norm(*v) for v in zip(x, m, sd)]).tobytes()
    cnt = np.floor(rng.uniform(0, 60, 5000))
    lam = rng.uniform(0.01, 50, 5000)
    assert A.device_eval(9, cnt, lam).tobytes() == np.array([L.orc_ld_pois(*v) for v in zip(cnt, lam)]).tobytes()
    th = rng.uniform(0, 1, 5000)
    assert A.device_eval(10, th, np.full(5000, 2.0), np.full(5000, 2.0)).tobytes() == \
        np.array([L.orc_ld_beta(v, 2, 2) for v in th]).tobytes()
    xb = np.floor(rng.uniform(0, 2, 5000))
    assert A.device_eval(11, xb, th).tobytes() == np.array([L.orc_ld_bern(*v) for v in zip(xb, th)]).tobytes()
    assert A.device_eval(12, x, np.zeros(5000), np.full(5000, 100.0)).tobytes() == \
        np.array([L.orc_ld_unif(v, 0, 100) for v in x]).tobytes()
    r = np.concatenate([rng.uniform(-50, 50, 5000), [0.5, -0.5, 1.5, -1.5, 2.5, 0.49999999999999994]])
    assert A.device_eval(7, r).tobytes() == np.array([L.orc_js_round(v) for v in r]).tobytes()


def test_device_philox_stream():
    n = 4096
    idx = np.arange(n, dtype=np.float64)
    got = A.device_eval(8, np.full(n, 20260925.0), np.full(n, 12345.0), idx)
    assert got.tobytes() == synth.uniforms(20260925, 12345, n).tobytes()


def test_device_ld_functions_and_pow_equal_the_reference_goldens():
    """Every scalar ld.* / helper and Math.pow evaluated ON THE DEVICE against the reference's recorded outputs."""
    import ctypes as C
    import os
    import golden_io
    L = A.selftest_lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "ld_values.bin"), dtype="<f8").reshape(-1, 6)
    rec = np.ascontiguousarray(a[:, :5])
    out = np.empty(a.shape[0])
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    assert L.amwg_ld_device(0, a.shape[0], dp(rec), dp(out)) == 0
    want = a[:, 5]
    bad = [(r.tolist(), w, g) for r, w, g in zip(rec, want, out) if not ((w != w and g != g) or np.float64(w).tobytes() == np.float64(g).tobytes())]
    assert not bad, bad[:3]
    p = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_pow_pairs.bin"), dtype="<f8").reshape(-1, 3)
    got = A.device_eval(13, p[:, 0], p[:, 1])
    ok = (got.view(np.uint64) == p[:, 2].view(np.uint64)) | (np.isnan(got) & np.isnan(p[:, 2]))
    assert ok.all(), p[~ok][:3]


def test_two_valued_sum_fast_forward_is_exact_including_ties():
    """csrc/amwg_models.h two_valued_sum against the plain loop on the device, bit for bit: random addends, and addends whose
    significands end in z zero bits, which tie (sit exactly half-way between two multiples of ulp(acc)) in the binade where
    acc is 2^(z+1) times larger -- for large z a long binade, each of the four tie cases (none / one with even or odd other
    increment / both), starting accumulators of either sign and magnitude."""
    import ctypes as C
    L = A.selftest_lib()
    rng = np.random.default_rng(11)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for n, p_one in ((100000, 0.3), (5000, 0.5), (4099, 0.02), (777, 0.97), (64, 0.5), (1, 0.5), (0, 0.5)):
        x = (rng.uniform(size=n) < p_one).astype(np.float64)
        m = 4096
        l1 = -np.exp(rng.uniform(-12, 3, m))
        l0 = -np.exp(rng.uniform(-12, 3, m))
        # clear z trailing bits of the significands (z differs per addend): ties in the binade 2^(z+1) above the addend
        z1, z0 = rng.integers(0, 30, m), rng.integers(0, 30, m)
        u1, u0 = l1.view(np.uint64).copy(), l0.view(np.uint64).copy()
        u1 = (u1 >> z1.astype(np.uint64)) << z1.astype(np.uint64)
        u0 = (u0 >> z0.astype(np.uint64)) << z0.astype(np.uint64)
        u1 |= (np.uint64(1) << z1.astype(np.uint64)) * (rng.integers(0, 2, m).astype(np.uint64))     # sometimes make bit z the lowest set bit
        u0 |= (np.uint64(1) << z0.astype(np.uint64)) * (rng.integers(0, 2, m).astype(np.uint64))
        l1, l0 = u1.view(np.float64).copy(), u0.view(np.float64).copy()
        l1[:8] = [-np.inf, np.nan, 0.0, -0.0, 1.5, -5e-324, -1e-310, -1.0]          # the term-by-term fallback cases
        l0[8:12] = [-np.inf, np.nan, 0.0, 2.0]
        l1[12:16], l0[12:16] = [-1.0, -0.5, -0.75, -3.0], [-0.5, -0.5, -0.25, -3.0]   # power-of-two-ish addends: many ties
        acc0 = np.where(rng.uniform(size=m) < 0.5, rng.normal(0, 3, m), -np.exp(rng.uniform(-5, 25, m)))
        acc0[16:20] = [0.0, -0.0, 1e300, -1e300]
        ff, seq = np.empty(m), np.empty(m)
        assert L.amwg_two_valued_sum_check(0, dp(np.ascontiguousarray(x)), n, m, dp(acc0), dp(l1), dp(l0), dp(ff), dp(seq)) == 0
        same = (ff.view(np.uint64) == seq.view(np.uint64)) | (np.isnan(ff) & np.isnan(seq))
        bad = np.where(~same)[0]
        assert bad.size == 0, (n, bad[:5], acc0[bad[:5]], l1[bad[:5]].view(np.uint64), l0[bad[:5]].view(np.uint64), ff[bad[:5]], seq[bad[:5]])


def test_device_log1p_expm1_equal_v8():
    import os
    import golden_io
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_log1p_expm1_pairs.bin"), dtype="<f8").reshape(-1, 3)
    for op, col in ((14, 1), (15, 2)):
        got = A.device_eval(op, a[:, 0])
        ok = (got.view(np.uint64) == a[:, col].view(np.uint64)) | (np.isnan(got) & np.isnan(a[:, col]))
        assert ok.all(), (op, a[~ok][:3], got[~ok][:3])


def test_device_tanh_atan_log10_equal_v8():
    import os
    import golden_io
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math2_pairs.bin"), dtype="<f8").reshape(-1, 4)
    for op, col, arg in ((16, 1, a[:, 0]), (17, 2, a[:, 0]), (18, 3, np.abs(a[:, 0]))):
        got = A.device_eval(op, arg)
        ok = (got.view(np.uint64) == a[:, col].view(np.uint64)) | (np.isnan(got) & np.isnan(a[:, col]))
        assert ok.all(), (op, a[~ok][:3], got[~ok][:3])


def test_device_trigonometric_hyperbolic_and_root_twins_equal_v8():
    """The same goldens as tests/test_core_host.py, evaluated on the device (amwg_device_eval op 20: y = function id)."""
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math3_pairs.bin"), dtype="<f8").reshape(-1, 15)
    cols = {"sin": (3, 3, 0), "cos": (4, 4, 0), "tan": (5, 5, 0), "sinh": (8, 6, 0), "cosh": (9, 7, 0), "asinh": (10, 8, 0), "cbrt": (13, 9, 0), "log2": (14, 10, 0),
            "asin": (6, 11, 1), "acos": (7, 12, 1), "atanh": (12, 13, 1), "acosh": (11, 14, 2)}
    for name, (fn, col, argc) in cols.items():
        arg = np.abs(a[:, argc]) if name == "log2" else a[:, argc]
        got = A.device_eval(20, np.ascontiguousarray(arg), np.full(len(arg), float(fn)))
        want = a[:, col]
        ok = (got.view(np.uint64) == np.ascontiguousarray(want).view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert ok.all(), (name, arg[~ok][:3], got[~ok][:3], want[~ok][:3])
    b = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_atan2_pairs.bin"), dtype="<f8").reshape(-1, 3)
    got = A.device_eval(21, np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1]))
    assert ((got.view(np.uint64) == np.ascontiguousarray(b[:, 2]).view(np.uint64)) | (np.isnan(got) & np.isnan(b[:, 2]))).all()
    h = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_hypot_pairs.bin"), dtype="<f8").reshape(-1, 5)
    g3 = A.device_eval(22, np.ascontiguousarray(h[:, 0]), np.ascontiguousarray(h[:, 1]), np.ascontiguousarray(h[:, 2]))
    g2 = A.device_eval(23, np.ascontiguousarray(h[:, 0]), np.ascontiguousarray(h[:, 1]))
    assert ((g3.view(np.uint64) == np.ascontiguousarray(h[:, 4]).view(np.uint64)) | (np.isnan(g3) & np.isnan(h[:, 4]))).all()
    assert ((g2.view(np.uint64) == np.ascontiguousarray(h[:, 3]).view(np.uint64)) | (np.isnan(g2) & np.isnan(h[:, 3]))).all()


def test_device_js_mod_and_toint32_equal_v8():
    """`%` (js_mod) and `x | 0` (js_toint32) ON THE DEVICE against V8's own results (tests/golden/v8_mod_pairs.bin,
    oracle/gen_mod_golden.js): -0 % 3 = -0, -6 % 3 = -0, 5.5 % 0.1, 1e20 % 3, subnormals, +-inf, NaN.  Round 1's device
    `%` was the toolchain's frem expansion and returned +0 for a -0 dividend on the GPU box."""
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_mod_pairs.bin"), dtype="<f8").reshape(-1, 4)
    x, y = np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1])
    for op, want in ((24, np.ascontiguousarray(a[:, 2])), (25, np.ascontiguousarray(a[:, 3]))):
        got = A.device_eval(op, x, y)
        ok = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert ok.all(), (op, x[~ok][:5], y[~ok][:5], got[~ok][:5], want[~ok][:5])


def test_device_fused_softplus_equals_full_flow():
    """log1p_exp_v8 (Math.log1p(Math.exp(eta)) of a logistic likelihood as one straight line: fdlibm's branches as selects, both quotients
    without the exponent juggling of the general division) on the device against log1p_v8(exp_v8_full(x)) on the device, and that against V8
    on the host's golden pairs -- incl. arguments whose exp() lands next to every threshold the selects replace (tests/host/softplus_fuzz.cpp)."""
    rng = np.random.default_rng(777)
    parts = [rng.uniform(-8, 8, 1_000_000), rng.uniform(-22, 38, 1_000_000), rng.uniform(-745.5, 710, 200_000),
             np.ldexp(rng.uniform(0.5, 1, 300_000), rng.integers(-60, 7, 300_000)) * rng.choice([-1.0, 1.0], 300_000)]
    for h in (0x3FDA827A, 0x3e200000, 0x43400000, 0x3ff00000, 0x3fe00000, 0x40000000):
        for d in range(-3, 4):
            v = ((np.uint64(h + d) << np.uint64(32)) | rng.integers(0, 2 ** 32, 20000, dtype=np.uint64)).view(np.float64)
            x = np.log(v)
            parts += [x, np.nextafter(x, 1e300), np.nextafter(x, -1e300)]
    for h in (0x3ff6a09e, 0x3ff00000, 0x3ffffffd, 0x3ff00004, 0x3ff80000):      # 1 + exp(x) = m 2^k, m next to sqrt(2), 1 and 2
        for d in range(-3, 4):
            m = ((np.uint64(h + d) << np.uint64(32)) | rng.integers(0, 2 ** 32, 40000, dtype=np.uint64)).view(np.float64)
            t = np.ldexp(m, rng.integers(0, 53, 40000)) - 1.0
            x = np.log(t[t > 0])
            parts += [x, np.nextafter(x, 1e300), np.nextafter(x, -1e300)]
    parts.append(np.array([-20.0, 36.0, np.nextafter(-20.0, -1e300), np.nextafter(36.0, 1e300), -20.10126823623841, 36.7368005696771, 30.0, 0.0, -0.0, 1.0, -1.0,
                           np.inf, -np.inf, np.nan, 709.782712893384, -745.1332191019411, 1e-300, 5e-324, 0.34657359027997264, -0.8813735870195429]))
    x = np.concatenate(parts)
    want = A.device_eval(34, x)
    same = lambda p, q: np.array_equal(p.view(np.uint64)[~np.isnan(q)], q.view(np.uint64)[~np.isnan(q)]) and np.array_equal(np.isnan(p), np.isnan(q))
    assert same(A.device_eval(32, x), want)
    assert same(A.device_eval(33, x), want)
    assert same(A.device_eval(35, x), want)      # the branch-free form + its flag
    # the yardstick itself against Node (Math.log1p(Math.exp(x)) of tests/golden/v8_softplus_pairs.bin, oracle/gen_math_pairs.js)
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_softplus_pairs.bin"), dtype="<f8").reshape(-1, 2)
    assert A.device_eval(34, a[:, 0]).tobytes() == np.ascontiguousarray(a[:, 1]).tobytes()
    assert A.device_eval(32, a[:, 0]).tobytes() == np.ascontiguousarray(a[:, 1]).tobytes()
