"""Pins the oracle's exp/log/Philox/ld.* pieces.

v8_math_pairs.bin = outputs of Node's own Math.exp/Math.log (oracle/gen_math_pairs.js);
Philox vectors are the Random123 known-answer tests; ld.* values are outputs of the
reference's distributions.js recorded in SURVEY.md Appendix B7.
"""
import ctypes as C
import os

import numpy as np

import golden_io
import oracle_lib
import synth


def test_exp_log_bit_exact_vs_v8():
    L = oracle_lib.lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert a.shape[0] == 120000
    bad_e = bad_l = 0
    for x, e, l in a:
        ge, gl = L.orc_exp(x), L.orc_log(abs(x))
        bad_e += np.float64(ge).tobytes() != np.float64(e).tobytes()
        bad_l += np.float64(gl).tobytes() != np.float64(l).tobytes()
    assert bad_e == 0 and bad_l == 0


KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    L = oracle_lib.lib()
    for ctr, key, want in KAT:
        c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        L.orc_philox4x32_10(c, k, o)
        assert tuple(o) == want
        got = synth.philox4x32_10(*[np.array([v], dtype=np.uint64) for v in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_uniform_stream_matches_numpy_twin():
    L = oracle_lib.lib()
    u = synth.uniforms(20260925, 7, 101)
    for i in range(101):
        assert L.orc_uniform(20260925, 7, i) == u[i]
    assert 0.0 <= u.min() and u.max() < 1.0


def test_ld_values_from_reference():
    L = oracle_lib.lib()
    # outputs of the reference's distributions.js
    assert L.orc_ld_norm(183, 184.4, 4.9) == -2.5489900648518664
    assert L.orc_ld_unif(5, 0, 100) == -4.605170185988091
    assert L.orc_ld_pois(3, 10) == -4.884004190245918
    assert L.orc_ld_pois(10, 10) == -2.0785616431359326
    assert L.orc_ld_beta(0.3, 2, 2) == 0.23111172096338728
    assert L.orc_ld_bern(0, 0.3) == -0.35667494393873245
    assert L.orc_ld_bern(1, 0.3) == -1.2039728043259361
    assert abs(L.orc_lgamma(10) - 12.801827480082) < 1e-11
    assert L.orc_ld_unif(-1, 0, 100) == float("-inf")
    assert L.orc_ld_beta(1.5, 2, 2) == float("-inf")
    assert L.orc_ld_beta(0.3, 1, 1) == 0.0
    assert L.orc_ld_bern(0.5, 0.3) == float("-inf")
    assert L.orc_ld_pois(-1, 3) == float("-inf")


def test_js_round_half_up():
    L = oracle_lib.lib()
    for x, w in [(0.5, 1.0), (-0.5, -0.0), (1.5, 2.0), (-1.5, -1.0), (2.4999, 2.0), (-2.5001, -3.0),
                 (0.49999999999999994, 0.0), (1e300, 1e300), (-7.0, -7.0), (-0.2, -0.0), (-0.0, -0.0), (0.2, 0.0), (4503599627370497.0, 4503599627370497.0)]:
        assert np.float64(L.orc_js_round(x)).tobytes() == np.float64(w).tobytes(), x      # bits: Math.round(-0.2) is -0


def _same(a, b):
    return (a != a and b != b) or np.float64(a).tobytes() == np.float64(b).tobytes()


def test_every_ld_function_and_pow_pinned_against_the_reference():
    """oracle/amwg_oracle.c orc_ld (all 22 scalar densities/helpers of distributions.js) against 13 200 outputs of the
    unmodified reference; oracle_math.h om_pow against 60 000 outputs of Node's Math.pow (oracle/gen_ld_golden.js)."""
    L = oracle_lib.lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "ld_values.bin"), dtype="<f8").reshape(-1, 6)
    assert a.shape[0] == 13200
    for r in a:
        assert _same(L.orc_ld(int(r[0]), r[1], r[2], r[3], r[4]), r[5]), r.tolist()
    p = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_pow_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert sum(not _same(L.orc_pow(x, y), w) for x, y, w in p) == 0


def test_log1p_expm1_pinned_against_v8():
    L = oracle_lib.lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_log1p_expm1_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert a.shape[0] == 60000
    assert sum((not _same(L.orc_log1p(x), l)) + (not _same(L.orc_expm1(x), e)) for x, l, e in a) == 0


def test_tanh_atan_log10_pinned_against_v8():
    L = oracle_lib.lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math2_pairs.bin"), dtype="<f8").reshape(-1, 4)
    assert a.shape[0] == 60000
    assert sum((not _same(L.orc_tanh(x), t)) + (not _same(L.orc_atan(x), at)) + (not _same(L.orc_log10(abs(x)), lg)) for x, t, at, lg in a) == 0
