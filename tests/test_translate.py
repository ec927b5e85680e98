"""The log_post translator (bayes.js_amd/translate.js) without a GPU.

For every closure of tests/js/user_models.js (the reference's own fixtures, tests/test_data.js, the README programs,
the BASELINE cfg4/cfg5 closures, and models covering the rest of distributions.js):
  * the generated HIP text, compiled for the HOST, returns bit for bit the value the closure returned under the
    reference's distributions.js on V8 (tests/golden/user_*.json, oracle/gen_user_golden.js), derived quantities included;
  * evaluating it in the order of G lanes per chain agrees with one lane to rounding;
  * hiprtc compiles it, together with the step kernel, for gfx950 (no device needed).
"""
import ctypes as C
import math
import os
import shutil

import numpy as np
import pytest

import amwg_ctypes as A
import golden_io
import user_host

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")

NAMES = ["readme_normal", "readme_bern", "norm_post_derived", "complex_model", "hier_binomial", "multi_bern", "multivar_poisson",
         "hier_normal_closure", "hier_rows_bounded", "hier_rows_int", "pois_glm_closure", "spike_slab", "survival_mix", "discrete_mix", "mixture_arrays", "many_named", "semantics_probe", "logistic_softplus", "modern_js", "live_out_temp", "circular_wrapped_cauchy", "structured_helpers", "records_logistic", "categorical_arms", "pois_const_rate", "binom_const_size", "logit_n10k", "logit_bern_n10k", "wide_regression", "long_dim", "undefined_reads", "readme_normal_swapped"] + ["cfgfuzz_%d" % k for k in range(16)]
BIG_SHAPES = ("wide_regression", "long_dim")     # 20 named parameters + 19 data arrays; dim [300]: short runs, fewer recorded states


def same(a, b):
    return (math.isnan(a) and math.isnan(b)) or np.float64(a).tobytes() == np.float64(b).tobytes()


@pytest.mark.parametrize("name", NAMES)
def test_translated_closure_equals_reference_on_host(name):
    gold = golden_io.load("user_" + name)
    m = user_host.host_model(name)
    assert len(gold["log_post_checks"]) >= (15 if name.startswith("cfgfuzz") else (10 if name in BIG_SHAPES else 30))
    finite = 0
    for chk in gold["log_post_checks"]:
        got, dv = m.eval(chk["state"], 1, derived=True)
        assert same(got, chk["log_post"]), (name, chk["state"], got, chk["log_post"])
        assert len(dv) == len(chk["derived"]) and all(same(a, b) for a, b in zip(dv, chk["derived"]))
        finite += math.isfinite(chk["log_post"])
    assert finite >= (5 if name.startswith("cfgfuzz") or name in BIG_SHAPES else 10)


@pytest.mark.parametrize("name", NAMES)
def test_lane_split_order_agrees_to_rounding(name):
    gold = golden_io.load("user_" + name)
    m = user_host.host_model(name)
    for chk in gold["log_post_checks"][:12]:
        one = m.eval(chk["state"], 1)
        for lanes in (2, 4, 64):
            if not m.meta["parallel"] and lanes > 1:
                continue
            v = m.eval(chk["state"], lanes)
            if math.isfinite(one):
                assert abs(v - one) <= 1e-11 * max(1.0, abs(one)), (name, lanes, v, one)
            else:
                assert same(v, one) or (math.isnan(v) and math.isnan(one))


def test_lane_split_flags():
    meta = {n: user_host.host_model(n).meta for n in NAMES}
    assert meta["readme_normal"]["parallel"] == 1 and meta["pois_glm_closure"]["parallel"] == 1
    assert meta["multi_bern"]["parallel"] == 0            # `return expr`: nothing to split
    assert meta["norm_post_derived"]["derived"] == ["var"]
    assert meta["readme_normal"]["lds_bytes"] == 16 and meta["readme_normal"]["array_types"] == [1]   # integer heights: u8 storage


@pytest.mark.parametrize("name", NAMES)
def test_hiprtc_compiles_for_gfx950(name):
    m = user_host.host_model(name)
    L = A.lib()
    n = C.c_size_t(0)
    lanes = 4 if m.meta["parallel"] else 1
    rc = L.amwg_compile_user(m.source.encode(), lanes, min(256, m.meta["max_threads"]), b"gfx950", C.byref(n))
    assert rc == 0, L.amwg_last_error().decode()[-3000:]
    assert n.value > 10000


def oracle_spec(name):
    """OracleChain spec for a user closure: the oracle's stepper + the host build of the translated closure as log_post."""
    gold = golden_io.load("user_" + name)
    rec = gold["chains"][0]
    m = user_host.host_model(name)
    params, init, opts = [], [], []
    for p in rec["params_completed"]:
        ln = int(np.prod(p["dim"]))
        params.append({"type": p["type"], "len": ln, "top": p["dim"][0], "multidim": 0 if p["dim"] == [1] else 1,
                       "lower": p["lower"], "upper": p["upper"]})
        init += p["init"]
    for o in rec["comp_opts"]:
        opts.append({"prop_log_scale": o.get("prop_log_scale", 0.0), "max_adaptation": o.get("max_adaptation", 0.33),
                     "initial_adaptation": o.get("initial_adaptation", 1.0), "target_accept_rate": o.get("target_accept_rate", 0.44),
                     "batch_size": o.get("batch_size", 50), "is_adapting": o.get("is_adapting", True)})
    return {"log_post_fn": lambda st, lanes: m.eval(st, lanes), "params": params, "P": len(init), "init": init, "comp_opts": opts}, gold, m


@pytest.mark.parametrize("name", ["complex_model", "spike_slab", "multi_bern", "hier_binomial", "discrete_mix", "modern_js", "multivar_poisson", "semantics_probe", "circular_wrapped_cauchy", "structured_helpers", "records_logistic", "categorical_arms"]
                         + ["cfgfuzz_%d" % k for k in range(16)] + ["cfgedge_%d" % k for k in range(10)])
def test_oracle_stepper_with_translated_closure_reproduces_reference(name):
    """Pins the oracle's BinaryStepper (mcmc.js:753-767) and int/real steppers on user models: the C oracle, stepping with the
    host build of the translated closure as log_post, reproduces the seeded reference run bit for bit.  cfgfuzz_*: randomly drawn
    sampler configurations (types, dims up to three levels, bounds, inits, global / per-parameter / per-component options, schedules
    with stop/start_adaptation and thinning; tests/js/user_models.js makeConfigCase)."""
    import oracle_lib
    spec, gold, m = oracle_spec(name)
    P = spec["P"]
    for rec in gold["chains"]:
        o = oracle_lib.OracleChain(spec, gold["case"]["seed"], rec["chain"], lanes=1)
        k = 0
        for seg in gold["case"]["schedule"]:
            if seg["op"] == "burn":
                o.burn(seg["n"])
            elif seg["op"] in ("stop", "start"):
                o.set_adapting(seg["op"] == "start")
            else:
                got = o.sample(seg["n"], seg.get("thin", 1))
                want = rec["samples"][k]
                k += 1
                if want["kept"] == 0:          # sample(0): an empty result, nothing consumed
                    assert got.shape[0] == 0
                    continue
                w = np.array(want["draws"], dtype=np.float64)[:, :P]
                assert got[: w.shape[0]].tobytes() == np.ascontiguousarray(w).tobytes()
        assert o.state().tolist() == rec["final_state"]
        info = o.info()
        assert info["accepts"].tolist() == rec["accepts"] and info["inbounds"].tolist() == rec["inbounds"]
        assert info["prop_log_scale"].tolist() == rec["prop_log_scale"]
        assert o.uniforms() == rec["uniforms"]


def test_hiprtc_errors_surface_with_the_compiler_log():
    L = A.lib()
    n = C.c_size_t(0)
    bad = b"namespace amwg { struct UserModel { static constexpr bool kUser = true; this is not C++ }; }"
    assert L.amwg_compile_user(bad, 1, 256, b"gfx950", C.byref(n)) == -1
    msg = L.amwg_last_error().decode()
    assert "did not compile" in msg and "error" in msg


def test_two_valued_sum_host_fuzz(tmp_path):
    """csrc/amwg_twoval.h compiled for the host: 200 000 random / adversarial (data, acc0, l1, l0) cases -- trailing-zero
    significands (ties), either sign of acc0, non-finite and positive addends -- against the plain fp64 loop, bit for bit."""
    import os
    import subprocess
    exe = str(tmp_path / "twoval_fuzz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "twoval_fuzz.cpp"), "-o", exe])
    p = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "mismatches=0" in p.stdout, p.stdout[-2000:]


def test_k_valued_sum_host_fuzz(tmp_path):
    """csrc/amwg_kval.h compiled for the host (K = 1, 2, 3, 5, 8, 16 distinct addends): ~100 000 random / adversarial (data, acc0, addends)
    cases -- trailing-zero significands (ties in reachable binades: summed term by term there), exact multiples of one another, either sign
    of acc0, non-finite and positive addends -- against the plain fp64 loop, bit for bit."""
    import os
    import subprocess
    exe = str(tmp_path / "kval_fuzz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "kval_fuzz.cpp"), "-o", exe])
    p = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "mismatches=0" in p.stdout, p.stdout[-2000:]


def test_constant_rate_count_loops_are_fast_forwarded_with_one_lane():
    """`lp += ld.pois(y[i], rate)` / `ld.binom(y[i], size, prob)` with loop-invariant parameters over small-integer data: the generated code
    evaluates the term once per distinct value and hands the loop to k_valued_sum when a chain has one lane (the reference's order); the
    lane-split loop stays for G > 1.  (That the result equals the reference's closure bit for bit is test_translated_closure_equals_reference_on_host.)"""
    import re
    for name, K in (("pois_const_rate", 13), ("binom_const_size", 8)):      # counts 0 .. 12; successes 0 .. 7
        m = user_host.host_model(name)
        found = re.search(r"kval_loop_one_lane<(\d+)>", m.source)
        assert found and int(found.group(1)) == K and "if constexpr (G == 1)" in m.source
        assert any(k.startswith("#aux:kval:") for k in m.meta["array_keys"])
        assert 0 < m.meta["work_one_lane"] < 0.05 * m.meta["work_per_eval"]
        # the tables are the one-lane plan's alone: the G > 1 plan stages the observations (one byte each) and the lfactorial column ld.pois reads,
        # not the K x N / 4 + N bytes of tables it never reads -- whose bytes would also feed the 73 728-byte rule of the workgroup limit
        n = len(m.arrays[0])
        assert m.meta["lds_bytes"] <= (n + 15 & ~15) + (8 * n + 15 & ~15)
        stage = m.source[m.source.index("static void stage("):m.source.index("template <int G, bool DERIVE>")]
        many = stage[stage.index("} else {"):]
        tabs = [j for j, k in enumerate(m.meta["array_keys"]) if k.startswith("#aux:kval:")]
        assert len(tabs) == 2 and not any("user_arr<%d>" % j in many for j in tabs) and any("user_arr<%d>" % j in stage for j in tabs)


def test_logistic_likelihoods_get_the_fused_softplus_and_wide_workgroups():
    """`y*eta - Math.log1p(Math.exp(eta))`: the generated code calls ONE device function for the softplus (csrc/amwg_math.h log1p_exp_v8), in the unrolled
    lane-split loop its branch-free form with one flag per block of terms and a re-evaluation through the full functions under that flag; a closure
    whose staged data leaves room for one workgroup per CU may use 512-thread workgroups (two wavefronts per SIMD), smaller ones keep 256.  (That the
    values equal the reference's bit for bit is test_translated_closure_equals_reference_on_host; the function itself: test_core_host.py, test_gpu_math.py.)"""
    m = user_host.host_model("logit_n10k")
    assert "log1p_exp_v8_open(rr_, v_eta)" in m.source and "if (rr_)" in m.source and "log1p_exp_cold(v_eta)" in m.source
    assert "log1p_v8(exp_v8(" not in m.source
    assert m.meta["max_threads"] == 512 and m.meta["lds_bytes"] > 73728
    b = user_host.host_model("logit_bern_n10k")
    assert "log1p_exp_v8" not in b.source and "ld_bern(" in b.source and b.meta["max_threads"] == 512
    small = user_host.host_model("records_logistic")
    assert "log1p_exp_v8" in small.source and small.meta["max_threads"] == 256


def test_division_by_invariant_host_fuzz(tmp_path):
    """csrc/amwg_div.h compiled for the host: 8 million quotients (divisor 2^-200..2^200, numerator 2^-600..2^600 or 0, all-ones and
    power-of-two significands among them) equal IEEE division bit for bit."""
    import os
    import subprocess
    exe = str(tmp_path / "div_fuzz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "div_fuzz.cpp"), "-o", exe])
    p = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "mismatches=0" in p.stdout, p.stdout[-2000:]


@pytest.mark.parametrize("seed", [3, 5])
def test_fuzzed_closures_equal_v8_on_host(seed):
    """Random closures (tests/js/fuzz_translate_cli.js: 53 derived quantities and an accumulated return value each, built from random
    arithmetic, comparisons, ?:, && ||, Math.*, ld.*, loops with if/else/continue/break, nested loops, local arrays, integer counters and
    index arithmetic, for-of / forEach / reduce / destructuring / arrow helpers, -0 / NaN / Infinity operands): the translator's text, compiled for the host, returns bit for bit what V8 returns at
    40 random states; evaluated in the order of 2, 4 and 64 lanes per chain the derived quantities stay identical and the sum agrees to
    rounding.  (This test found Math.round's -0; campaigns over some 60 further seeds are clean.)"""
    checked = 0
    for name in user_host.fuzz_models(seed, 2, 24):      # 24 derived quantities per program here (tools/fuzz_campaign.py runs the 48-quantity ones)
        m = user_host.host_model(name)
        for t, pt in enumerate(user_host.stepper_states(name)):
            state = [float(np.frombuffer(bytes.fromhex(h), dtype=">f8")[0]) for h in pt["state"]]
            got, dv = m.eval(state, 1, derived=True)
            want = [float(np.frombuffer(bytes.fromhex(h), dtype=">f8")[0]) for h in pt["derived"] + [pt["lp"]]]
            for key, a, b in zip(m.meta["derived"] + ["return"], dv + [got], want):
                assert same(a, b), (name, key, state, a, b)
                checked += 1
            if m.meta["parallel"] and t < 12:
                for lanes in (2, 4, 64):
                    v, dvl = m.eval(state, lanes, derived=True)
                    assert all(same(a, b) for a, b in zip(dvl, dv)), (name, lanes, state)
                    assert same(v, got) or (math.isfinite(got) and abs(v - got) <= 1e-9 * max(1.0, abs(got))), (name, lanes, state, v, got)
    assert checked >= 2 * 40 * 26


def test_compiled_closures_are_cached_on_disk(tmp_path, monkeypatch):
    """The second compilation of the same closure + geometry + target is a file read: a fresh process would load the code object instead of
    spending ~0.6 s in hiprtc (amwg_code_cache_stats counts this process's hits and misses; a changed geometry is a different key)."""
    monkeypatch.setenv("AMWG_CACHE_DIR", str(tmp_path / "cache"))
    m = user_host.host_model("readme_normal")
    L = A.lib()
    n = C.c_size_t(0)
    h0, m0, d = A.code_cache_stats()
    assert d == str(tmp_path / "cache")
    assert L.amwg_compile_user(m.source.encode(), 4, 256, b"gfx950", C.byref(n)) == 0
    first = n.value
    files = list((tmp_path / "cache").glob("*.hsaco"))
    assert len(files) == 1 and files[0].stat().st_size > first
    assert L.amwg_compile_user(m.source.encode(), 4, 256, b"gfx950", C.byref(n)) == 0 and n.value == first
    h1, m1, _ = A.code_cache_stats()
    assert (h1 - h0, m1 - m0) == (1, 1)
    assert L.amwg_compile_user(m.source.encode(), 8, 256, b"gfx950", C.byref(n)) == 0            # another geometry: another entry
    assert len(list((tmp_path / "cache").glob("*.hsaco"))) == 2
    files[0].write_bytes(files[0].read_bytes()[:100])                                              # a damaged file is ignored and replaced
    assert L.amwg_compile_user(m.source.encode(), 4, 256, b"gfx950", C.byref(n)) == 0 and n.value == first
    monkeypatch.setenv("AMWG_CACHE_DIR", "")                                                        # off
    assert A.code_cache_stats()[2] == ""
    assert L.amwg_compile_user(m.source.encode(), 4, 256, b"gfx950", C.byref(n)) == 0


def test_row_plan_is_found_proved_and_refused_where_it_must_be():
    """csrc/amwg_rows.h: a closure that ENDS in `lp += ld.norm(y[i], theta[g[i]], sd)` over all observations gets a row plan (head + the loop's three numbers);
    the sweep prefetch additionally needs the translator's proof that a lane's head reads theta only as its own entry.  Closures that do not end in such a
    loop, or whose head returns early / reads theta[const], get no plan / no sweep."""
    m = user_host.host_model("hier_normal_closure")
    assert (m.meta["rows_n_obs"], m.meta["rows_groups"], m.meta["rows_sweep"]) == (640, 8, 1)
    assert "kRowBase = 0, kRowGroups = 8" in m.source and "kRowSweep = true" in m.source and ": UserRows<UserModel>" in m.source
    head = m.source[m.source.index("static double head("):m.source.index("template <int G, bool DERIVE>")]
    assert "norm_data_loop_gather" not in head and "ld_norm_fast(S(v_k), S(8), k0" in head and "return v_lp;" in head
    b = user_host.host_model("hier_rows_bounded")      # theta is the THIRD parameter (state offset 2), the head has a hyper-parameter of its own
    assert (b.meta["rows_n_obs"], b.meta["rows_groups"], b.meta["rows_sweep"]) == (640, 8, 1) and "kRowBase = 2" in b.source and "return S(10);" in b.source
    assert user_host.host_model("hier_rows_int").meta["rows_sweep"] == 1
    for name in ("readme_normal", "pois_glm_closure", "hier_binomial", "logit_n10k"):      # no gathered normal loop at the end
        q = user_host.host_model(name)
        assert q.meta["rows_n_obs"] == 0 and "UserRows" not in q.source


def test_row_plan_proof_refuses_heads_that_read_other_entries(tmp_path):
    """the same likelihood with heads the proof must refuse: theta[0] read by a constant index (lane 0's sum then depends on TWO entries when its group is
    not 0 ... and on entry 0 for every lane's start), an early return, labels that do not start 0, 1, 2, ..."""
    import json
    import subprocess
    js = r"""
const t = require(process.argv[2]); const synth = require(process.argv[3]);
global.ld = require(process.argv[4]);
const d = synth.hier(640, 8, 20260925);
const P = { theta: { type: 'real', dim: [8], lower: -Infinity, upper: Infinity, init: [0.5,0.5,0.5,0.5,0.5,0.5,0.5,0.5] }, mu: { type: 'real', dim: [1], lower: -Infinity, upper: Infinity, init: 0.5 }, sigma: { type: 'real', dim: [1], lower: 0, upper: Infinity, init: 1 } };
const lik = 'for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma); return lp; }';
const out = {};
const run = (k, src, data) => { const r = t.translate(src, P, data || d, {}); out[k] = [r.rows_n_obs, r.rows_groups, r.rows_sweep, /kRowSweep = (true|false)/.exec(r.source) ? RegExp.$1 : null]; };
run('plain', 'function (s, d) { let lp = 0; for (let k = 0; k < 8; k++) lp += ld.norm(s.theta[k], s.mu, 10); ' + lik);
run('const_index', 'function (s, d) { let lp = 0; lp += ld.norm(s.theta[0], s.mu, 10); ' + lik);
run('early_return', 'function (s, d) { let lp = 0; if (s.sigma > 50) return -Infinity; ' + lik);
run('partial_loop', 'function (s, d) { let lp = 0; for (let k = 0; k < 4; k++) lp += ld.norm(s.theta[k], s.mu, 10); ' + lik);
run('shifted_labels', 'function (s, d) { let lp = 0; for (let k = 0; k < 8; k++) lp += ld.norm(s.theta[k], s.mu, 10); ' + lik, Object.assign({}, d, { g: d.g.map((v) => (v + 1) % 8) }));
run('not_last', 'function (s, d) { let lp = 0; for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], s.sigma); lp += ld.norm(s.mu, 0, 100); return lp; }');
run('sd_local', 'function (s, d) { let lp = 0; const sd = Math.sqrt(s.sigma); for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], sd); return lp; }');
// (round-5 advisor finding: both of these came back "proved" -- the scan only looked for `S(` reads in the head)
const d2 = Object.assign({}, d, { z: d.y.map((v) => v + 1), g2: d.y.map((v, i) => (3 * i + 1) % 8), w: d.y.map((v) => v - 1), h: d.y.map((v, i) => i % 4) });
const pri = 'function (s, d) { let lp = 0; for (let k = 0; k < 8; k++) lp += ld.norm(s.theta[k], s.mu, 10); ';
run('earlier_gather', pri + 'for (let i = 0; i < d.z.length; i++) lp += ld.norm(d.z[i], s.theta[d.g2[i]], s.sigma); ' + lik, d2);
run('sd_reads_theta', pri + 'for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], Math.abs(s.theta[0]) + 1); return lp; }');
run('sd_reads_mu', pri + 'for (let i = 0; i < d.y.length; i++) lp += ld.norm(d.y[i], s.theta[d.g[i]], Math.abs(s.mu) + 1); return lp; }');
const P2 = Object.assign({ phi: { type: 'real', dim: [4], lower: -Infinity, upper: Infinity, init: [0, 0, 0, 0] } }, P);
const r2 = t.translate(pri + 'for (let i = 0; i < d.w.length; i++) lp += ld.norm(d.w[i], s.phi[d.h[i]], 3); ' + lik, P2, d2, {});      // an earlier gathered loop over ANOTHER vector
out.earlier_gather_other_vector = [r2.rows_n_obs, r2.rows_groups, r2.rows_sweep, /kRowSweep = (true|false)/.exec(r2.source) ? RegExp.$1 : null, r2.source.slice(r2.source.indexOf('static double head('), r2.source.indexOf('template <int G, bool DERIVE>')).indexOf('(A0, A1, S, 0, 4, 640,') > 0];
console.log(JSON.stringify(out));
"""
    f = tmp_path / "probe.js"
    f.write_text(js)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["node", str(f), os.path.join(root, "bayes.js_amd", "translate.js"), os.path.join(root, "oracle", "synth.js"), os.path.join(root, "bayes.js_amd", "ld.js")],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["plain"] == [640, 8, 1, "true"]
    assert out["const_index"] == [640, 8, 0, "false"] and out["partial_loop"] == [640, 8, 0, "false"] and out["shifted_labels"] == [640, 8, 0, "false"]      # lane reuse yes, sweep no
    assert out["early_return"][:3] == [0, 0, 0] and out["not_last"][:3] == [0, 0, 0] and out["sd_local"][:3] == [0, 0, 0]
    # the state handed on as a whole (an earlier gathered loop over the swept vector through other labels) and an sd that reads an entry of the vector: the
    # lanes' three numbers then depend on entries other than their own -- lane reuse yes, sweep no; the same shapes off the vector keep the proof
    assert out["earlier_gather"] == [640, 8, 0, "false"] and out["sd_reads_theta"] == [640, 8, 0, "false"]
    assert out["sd_reads_mu"] == [640, 8, 1, "true"] and out["earlier_gather_other_vector"] == [640, 8, 1, "true", True]


def test_poisson_tail_plan_is_found_and_refused_where_it_must_be(tmp_path):
    """translate.js poisTailPlan (csrc/amwg_ptail.h): a closure that ENDS in `lp += ld.pois(y[i], Math.exp(eta))` over all observations gets certified values at 16 lanes
    per chain.  Scalar-register state only if every state index is the same for all observations, the row cache only if every data read is of the observation's own row;
    no plan at all for a negative count (the reference's term is -inf), a loop that is not the last statement, a head that does more than add, an early return."""
    import ctypes as C
    import json
    import subprocess
    m = user_host.host_model("pois_glm_closure")
    assert m.meta["pois_tail_n"] == 500 and m.meta["cert_tail_n"] == 0 and m.meta["rows_cert"] == 0
    for token in ("kPoisTail = true, kCertified = true, kReferenceOrder = true", "kCertifiedLanes = 16, kTailN = 500, kStateN = 9", "kTailUniformState = true", "kTailRows = true", "kTailLinear = true", "kTailLinearRoundings = 23", "__builtin_fma(R.a1[(v___b1_k) - 0], S(v___b1_k), v_eta)",
                  "struct TailRow { uint8_t a0[1]; double a1[7]; };", "R.a1[(v___b1_k) - 0] * S(v___b1_k)", "pois_tail_approx<UserModel, G, BT>", "pois_tail_reference<UserModel, G>"):
        assert token in m.source, token
    L = A.lib()
    n = C.c_size_t(0)
    assert L.amwg_compile_user(m.source.encode(), 16, 256, b"gfx950", C.byref(n)) == 0, L.amwg_last_error().decode()[-3000:]
    js = r"""
const t = require(process.argv[2]); const synth = require(process.argv[3]);
global.ld = require(process.argv[4]);
const d = synth.glm(500, 20260925);
const P = { beta: { type: 'real', dim: [8], lower: -Infinity, upper: Infinity, init: [0,0,0,0,0,0,0,0] }, cp: { type: 'int', dim: [1], lower: 0, upper: 499, init: 250 } };
const pri = 'function (s, d) { let lp = 0; const N = d.y.length, K = d.K; for (let k = 0; k < 8; k++) lp += ld.norm(s.beta[k], 0, 10); lp += ld.unif(s.cp, 0, N - 1); ';
const loop = (eta) => 'for (let i = 0; i < N; i++) { let eta = 0; ' + eta + ' lp += ld.pois(d.y[i], Math.exp(eta)); } ';
const lin = 'for (let k = 0; k < K; k++) eta += d.X[i * K + k] * s.beta[k]; if (i >= s.cp) eta += s.beta[7];';
const out = {}, src = {};
const flag = (r, k) => { const m = new RegExp(k + ' = (true|false)').exec(r.source); return m ? m[1] : null; };
const run = (k, text, data) => { const r = t.translate(text, P, data || d, {}); out[k] = [r.pois_tail_n, flag(r, 'kTailUniformState'), flag(r, 'kTailRows'), flag(r, 'kTailLinear')]; src[k] = r.source; };
run('plain', pri + loop(lin) + 'return lp; }');
run('negative_count', pri + loop(lin) + 'return lp; }', Object.assign({}, d, { y: d.y.map((v, i) => (i === 7 ? -1 : v)) }));
run('not_last', pri + loop(lin) + 'lp += ld.norm(s.beta[0], 0, 1); return lp; }');
run('early_return', 'function (s, d) { let lp = 0; const N = d.y.length, K = d.K; if (s.beta[0] > 50) return -Infinity; ' + loop(lin) + 'return lp; }');
run('head_scales', pri + 'lp = lp * 0.5; ' + loop(lin) + 'return lp; }');
run('gathered_state', pri + loop('eta = s.beta[d.y[i] % 8] + d.X[i * K] * s.beta[1];') + 'return lp; }');
run('next_row', pri + loop('eta = d.X[((i + 1) % N) * K] * s.beta[0];') + 'return lp; }');
run('minus_and_literal', pri + loop('eta = 0.25; eta -= d.X[i * K + 1] * s.beta[1]; eta += s.beta[2] * d.X[i * K + 3]; eta -= s.beta[7];') + 'return lp; }');
run('product_of_states', pri + loop('eta = d.X[i * K] * s.beta[0] * s.beta[1];') + 'return lp; }');
run('eta_in_a_condition', pri + loop(lin + ' if (eta > 3) eta += s.beta[6];') + 'return lp; }');
out.switched_off = ((r) => [r.pois_tail_n, /kPoisTail/.test(r.source)])(t.translate(pri + loop(lin) + 'return lp; }', P, d, { no_pois_tail: true }));
require('fs').writeFileSync(process.argv[5], JSON.stringify(src));
console.log(JSON.stringify(out));
"""
    f = tmp_path / "probe.js"
    f.write_text(js)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["node", str(f), os.path.join(root, "bayes.js_amd", "translate.js"), os.path.join(root, "oracle", "synth.js"), os.path.join(root, "bayes.js_amd", "ld.js"), str(tmp_path / "src.json")],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["plain"] == [500, "true", "true", "true"]      # scalar-register state, row cache, linear predictor (fused steps, H from column maxima)
    assert out["minus_and_literal"] == [500, "true", "true", "true"] and out["product_of_states"] == [500, "true", "true", "false"] and out["eta_in_a_condition"] == [500, "true", "true", "false"]
    assert out["negative_count"][0] == 0 and out["not_last"][0] == 0 and out["early_return"][0] == 0 and out["head_scales"][0] == 0 and out["switched_off"] == [0, False]
    assert out["gathered_state"] == [500, "false", "false", "false"]      # per-lane LDS reads of the state, the plain loop
    assert out["next_row"] == [500, "true", "false", "false"]            # scalar-register state, but a read that is not of the observation's own row: no row cache
    srcs = json.load(open(tmp_path / "src.json"))
    for k in ("gathered_state", "next_row", "minus_and_literal", "product_of_states"):      # the fallback paths (and a linear predictor with signs and a literal) compile too
        assert L.amwg_compile_user(srcs[k].encode(), 16, 256, b"gfx950", C.byref(n)) == 0, (k, L.amwg_last_error().decode()[-3000:])


def test_constant_norm_inv_is_folded_to_the_device_functions_bits(tmp_path):
    """translate.js foldConstantNormInv (round 6): `norm_inv(<literal>)` -- the loop invariants of ld.norm with a constant sd: V8's logarithm, two products, the
    correctly rounded reciprocal and its low word from the exact residual (BigInt) -- becomes literals in the generated source; each must be the bits
    csrc/amwg_user.h norm_inv() computes (tests/host/norm_inv_fold.cpp, 315 values of sd over 26 decades), and the README-style closures carry the folded form."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "norm_inv_fold")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "bayes.js_amd", "csrc"),
                           os.path.join(root, "tests", "host", "norm_inv_fold.cpp"), "-o", exe])
    js = subprocess.run(["node", os.path.join(root, "tests", "js", "norm_inv_fold_cli.js")], capture_output=True, text=True, timeout=120)
    assert js.returncode == 0, js.stderr[-1000:]
    p = subprocess.run([exe], input=js.stdout, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "checked=315 unfolded=0 mismatches=0" in p.stdout, p.stdout[-1500:]
    m = user_host.host_model("hier_normal_closure")
    assert "/* norm_inv(10.0) */" in m.source and "norm_inv(10.0);" not in m.source
