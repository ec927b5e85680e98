"""CPU-side checks of libamwg.so: it loads, exports the whole C ABI, its host build of the
kernel arithmetic is bit-identical to V8, and it refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import amwg_ctypes
import golden_io
import model_spec
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = amwg_ctypes.lib()
    hdr = open(os.path.join(ROOT, "include", "amwg.h")).read()
    declared = set(re.findall(r"\b(amwg_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(amwg_ctypes.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.amwg_version()
    # the building blocks exported one by one live in the test build only, not in the product library
    T = amwg_ctypes.selftest_lib()
    hdr = open(os.path.join(ROOT, "include", "amwg_selftest.h")).read()
    declared_t = set(re.findall(r"\b(amwg_[a-z_0-9]+)\s*\(", hdr))
    # (amwg_audit_fetch: the bound-audit build only -- libamwg_audit.so, -DAMWG_AUDIT --, in neither of these two libraries: tests/test_gpu_bound_audit.py checks it)
    assert "amwg_audit_fetch" in declared_t and not hasattr(L, "amwg_audit_fetch") and not hasattr(T, "amwg_audit_fetch")
    declared_t.discard("amwg_audit_fetch")
    assert declared_t == set(amwg_ctypes.SELFTEST_EXPORTS)
    for name in declared_t:
        assert getattr(T, name) is not None
        assert not hasattr(L, name), name + " leaked into the product library"


def test_host_math_bit_exact_vs_v8():
    L = amwg_ctypes.lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math_pairs.bin"), dtype="<f8").reshape(-1, 3)
    bad = 0
    for x, e, l in a:
        bad += np.float64(L.amwg_exp(x)).tobytes() != np.float64(e).tobytes()
        bad += np.float64(L.amwg_log(abs(x))).tobytes() != np.float64(l).tobytes()
    assert bad == 0


def test_host_uniform_stream_matches_twin():
    L = amwg_ctypes.lib()
    u = synth.uniforms(20260925, 3, 64)
    for i in range(64):
        assert L.amwg_uniform(20260925, 3, i) == u[i]
    u = synth.uniforms(7, (1 << 32) + 5, 4)      # 64-bit chain ids
    assert L.amwg_uniform(7, (1 << 32) + 5, 3) == u[3]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback_without_gpu():
    data = synth.normal(100, 1)
    spec = model_spec.build_spec("normal", data)
    with pytest.raises(amwg_ctypes.AmwgError) as ei:
        amwg_ctypes.Sampler(spec, chains=4, seed=1)
    assert "HIP" in str(ei.value) or "device" in str(ei.value)


def test_bad_arguments_are_rejected_before_touching_the_device():
    data = synth.normal(10, 1)
    spec = model_spec.build_spec("normal", data)
    with pytest.raises(amwg_ctypes.AmwgError):
        amwg_ctypes.Sampler(spec, chains=0, seed=1)
    with pytest.raises(amwg_ctypes.AmwgError):
        amwg_ctypes.Sampler(spec, chains=4, seed=1, lanes_per_chain=3)
    bad = model_spec.build_spec("normal", data)
    bad["model"] = "beta_bern"       # two params handed to a one-param model
    with pytest.raises(amwg_ctypes.AmwgError):
        amwg_ctypes.Sampler(bad, chains=4, seed=1)


def _same(a, b):
    return (a != a and b != b) or np.float64(a).tobytes() == np.float64(b).tobytes()


def test_host_build_of_every_ld_function_and_pow_equals_the_reference():
    """The kernel's own source (csrc/amwg_ld.h, amwg_math.h), compiled for the host: all 22 scalar densities/helpers of
    distributions.js on 13 200 seeded argument sets recorded from the unmodified reference (oracle/gen_ld_golden.js), and
    Math.pow on 60 000 pairs recorded from Node's V8."""
    L = amwg_ctypes.selftest_lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "ld_values.bin"), dtype="<f8").reshape(-1, 6)
    assert a.shape[0] == 13200 and set(a[:, 0].astype(int)) == set(range(22))
    for r in a:
        assert _same(L.amwg_ld_host(int(r[0]), r[1], r[2], r[3], r[4]), r[5]), r.tolist()
    p = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_pow_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert p.shape[0] == 60000
    assert sum(not _same(L.amwg_pow(x, y), w) for x, y, w in p) == 0


def test_host_build_of_log1p_expm1_equals_v8():
    L = amwg_ctypes.selftest_lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_log1p_expm1_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert sum((not _same(L.amwg_log1p(x), l)) + (not _same(L.amwg_expm1(x), e)) for x, l, e in a) == 0


def test_host_build_of_tanh_atan_log10_equals_v8():
    L = amwg_ctypes.selftest_lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math2_pairs.bin"), dtype="<f8").reshape(-1, 4)
    assert sum((not _same(L.amwg_math1(0, x), t)) + (not _same(L.amwg_math1(1, x), at)) + (not _same(L.amwg_math1(2, abs(x)), lg)) for x, t, at, lg in a) == 0


def test_host_build_of_the_trigonometric_hyperbolic_and_root_twins_equals_v8():
    """csrc/amwg_trig.h (sin cos tan asin acos atan2 sinh cosh asinh acosh atanh cbrt log2 hypot) against 24 000 outputs each of this
    Node's V8 (oracle/gen_math3_golden.js), arguments up to 1e300 (Payne-Hanek reduction), subnormals, +-0, NaN, infinities."""
    L = amwg_ctypes.selftest_lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_math3_pairs.bin"), dtype="<f8").reshape(-1, 15)
    cols = {"sin": (3, 3, 0), "cos": (4, 4, 0), "tan": (5, 5, 0), "sinh": (8, 6, 0), "cosh": (9, 7, 0), "asinh": (10, 8, 0), "cbrt": (13, 9, 0), "log2": (14, 10, 0),
            "asin": (6, 11, 1), "acos": (7, 12, 1), "atanh": (12, 13, 1), "acosh": (11, 14, 2)}
    for name, (fn, col, argc) in cols.items():
        bad = sum(not _same(L.amwg_math1(fn, abs(r[argc]) if name == "log2" else r[argc]), r[col]) for r in a)
        assert bad == 0, name
    b = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_atan2_pairs.bin"), dtype="<f8").reshape(-1, 3)
    assert sum(not _same(L.amwg_math2(0, y, x), w) for y, x, w in b) == 0
    h = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_hypot_pairs.bin"), dtype="<f8").reshape(-1, 5)
    assert sum((not _same(L.amwg_math2(1, p, q), h2)) + (not _same(L.amwg_hypot3(p, q, r), h3)) for p, q, r, h2, h3 in h) == 0


def test_host_js_mod_and_toint32_equal_v8():
    """`%` and `x | 0` of translated closures (csrc/amwg_user.h js_mod / js_toint32, host build of the same header the device
    compiles) against 40 000 pairs recorded from V8 (oracle/gen_mod_golden.js): every +-0 / inf / NaN combination, exact
    multiples (sign of a zero result = sign of the dividend), subnormals, exponent gaps of thousands of bits."""
    L = amwg_ctypes.selftest_lib()
    a = np.fromfile(os.path.join(golden_io.GOLDEN, "v8_mod_pairs.bin"), dtype="<f8").reshape(-1, 4)
    bad = [(x, y, w, L.amwg_math2(2, x, y)) for x, y, w, _ in a if not _same(L.amwg_math2(2, x, y), w)]
    assert not bad, bad[:5]
    bad = [(x, w, L.amwg_math2(3, x, 0.0)) for x, _, _, w in a if not _same(L.amwg_math2(3, x, 0.0), w)]
    assert not bad, bad[:5]


def test_fused_exp_log_host_fuzz(tmp_path):
    """csrc/amwg_math.h compiled for the host: the straight-line exp_v8 (one formula for k) and the fused exp_log_v8 of the Poisson pass
    equal the full fdlibm control flow bit for bit on ~12 million arguments: random ones, every high word next to the thresholds the
    shortcuts replace, and arguments whose exp() lands next to log's significand thresholds (tests/host/explog_fuzz.cpp; it also checks
    that both sides of each sliver were actually visited)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "explog_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "explog_fuzz.cpp"), "-o", exe])
    p = subprocess.run([exe, "600000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "mismatches=0" in p.stdout, p.stdout[-2000:]


def test_fused_softplus_host_fuzz(tmp_path):
    """csrc/amwg_math.h compiled for the host: log1p_exp_v8 (Math.log1p(Math.exp(eta)) of a logistic likelihood as one straight line of selects)
    equals log1p_v8(exp_v8_full(x)) -- fdlibm's full control flow -- bit for bit on ~12 million arguments, and Node's own
    Math.log1p(Math.exp(x)) on the 100 000 pairs of tests/golden/v8_softplus_pairs.bin (tests/host/softplus_fuzz.cpp; it also checks that every
    form of log1p was visited)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "softplus_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "softplus_fuzz.cpp"), "-o", exe])
    p = subprocess.run([exe, "600000", os.path.join(root, "tests", "golden", "v8_softplus_pairs.bin")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "mismatches=0" in p.stdout and "v8_mismatches=0" in p.stdout, p.stdout[-2000:]


def test_certified_bounds_host_replay_in_quad_precision(tmp_path):
    """The three derivations behind the certified decisions (csrc/amwg_models.h: Normal at one lane, the hierarchical family's row layout, the Poisson family at 16 lanes),
    replayed on the host: E (the reference's term-by-term running sum, fp64) and A (the kernels' cheaper form in the kernels' summation order, fp64) against the REAL
    number R in __float128 from the same inputs -- ordinary states and the device audit's adversarial ones (n in {1, 2, 17, 63, 65}, data at 1e8, sigma from 1e-6 to
    1e6, constant and tiny data, H next to 690, counts of 1e6).  Each half of a bound must hold as written in the header's comments (|E - R| <= bE, |A - R| <= bA)
    and |A - E| <= eps / 2.  The device side of the same audit: tools/bound_audit.py, tests/test_gpu_bound_audit.py (round-5 review, item 1)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bound_replay")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "bound_replay.cpp"), "-o", exe, "-lquadmath"])
    p = subprocess.run([exe, "25"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "bounds_hold=1" in p.stdout and "VIOLATION" not in p.stdout and "pieces_over_bound=0" in p.stdout, p.stdout[-2000:]      # (the pieces of every derivation add up to no more than the bound handed on)


def test_copy_out_prefaulter_touches_without_changing_a_byte():
    """csrc/amwg_core.hip Prefaulter (round 6: what makes sample()'s 1 GB copy-out run at the link's rate instead of the page-fault rate): on fresh, on
    already-resident and on oddly aligned buffers, with 0 / 1 / 4 helper threads and more chunks than pieces, every byte stays what it was."""
    import ctypes as C
    import numpy as np
    T = amwg_ctypes.selftest_lib()
    rng = np.random.default_rng(5)
    for nbytes, off, chunks, threads in ((1, 0, 1, 0), (4095, 1, 3, 1), (40 << 20, 0, 7, 4), (33 << 20, 13, 64, 4), (9 << 20, 4095, 2, 0)):
        raw = np.zeros(nbytes + off + 16, dtype=np.uint8)      # (fresh pages: calloc-like)
        view = raw[off:off + nbytes]
        assert T.amwg_prefault_selftest(C.c_void_p(view.ctypes.data), nbytes, chunks, threads) == 0
        assert not view.any()
        view[:] = rng.integers(0, 256, nbytes, dtype=np.uint8)
        want = view.copy()
        assert T.amwg_prefault_selftest(C.c_void_p(view.ctypes.data), nbytes, chunks, threads) == 0
        assert np.array_equal(view, want)
    assert T.amwg_prefault_selftest(None, 10, 1, 0) != 0
