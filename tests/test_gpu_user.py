"""-m gpu: translated closures through the C ABI (amwg_create_user).

* one lane per chain: whole trajectories vs the reference goldens are checked from the JS side (tests/js/test_gpu_user.js);
  here the cached log_post of many chains is compared with the HOST build of the same generated text;
* G lanes per chain: the device's lane-split sum equals the host emulation of the same order bit for bit, at the
  state every sampled chain ended in, and the chains still sample the right posterior.
"""
import shutil

import os

import numpy as np
import pytest

import amwg_ctypes as A
import golden_io
import user_host

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")]

INF = float("inf")


def spec_for(name):
    """Sampler spec from the golden's completed params (what the reference built) + the translated model."""
    gold = golden_io.load("user_" + name)
    rec = gold["chains"][0]
    m = user_host.host_model(name)
    params, init, opts = [], [], []
    for p, in zip(rec["params_completed"]):
        ln = int(np.prod(p["dim"]))
        params.append({"type": p["type"], "len": ln, "top": p["dim"][0], "multidim": 0 if p["dim"] == [1] else 1,
                       "lower": p["lower"], "upper": p["upper"]})
        init += p["init"]
    for o in rec["comp_opts"]:
        opts.append({"prop_log_scale": o.get("prop_log_scale", 0.0), "max_adaptation": o.get("max_adaptation", 0.33),
                     "initial_adaptation": o.get("initial_adaptation", 1.0), "target_accept_rate": o.get("target_accept_rate", 0.44),
                     "batch_size": o.get("batch_size", 50), "is_adapting": o.get("is_adapting", True)})
    user = user_host.user_spec_part(m.source, m.arrays, m.meta)
    return {"user": user, "params": params, "P": len(init), "init": init, "comp_opts": opts}, m, gold


@pytest.mark.parametrize("name,lanes", [("readme_normal", 1), ("readme_normal", 4), ("norm_post_derived", 8), ("complex_model", 2),
                                        ("hier_binomial", 2), ("hier_normal_closure", 16), ("pois_glm_closure", 4), ("pois_glm_closure", 64),
                                        ("hier_normal_closure", 64), ("hier_rows_bounded", 64), ("hier_rows_int", 64), ("hier_rows_bounded", 8),      # 64 lanes: the row plan (csrc/amwg_rows.h), swept / bounded / integer
                                        ("spike_slab", 4), ("survival_mix", 2), ("discrete_mix", 1), ("multi_bern", 1), ("multivar_poisson", 1), ("mixture_arrays", 2), ("many_named", 4), ("readme_bern", 1), ("readme_bern", 4), ("semantics_probe", 1), ("logistic_softplus", 4), ("modern_js", 1), ("live_out_temp", 4), ("circular_wrapped_cauchy", 8), ("structured_helpers", 4), ("records_logistic", 4), ("categorical_arms", 2), ("pois_const_rate", 1), ("pois_const_rate", 4), ("binom_const_size", 1), ("binom_const_size", 8), ("logit_n10k", 1), ("logit_n10k", 64), ("logit_bern_n10k", 16),     # K-valued fast-forward with one lane; the split loop otherwise
                                        ("wide_regression", 1), ("wide_regression", 4), ("long_dim", 1), ("long_dim", 4)])     # > 16 named parameters, > 16 data arrays, dim [300]
def test_device_lane_sum_equals_host_emulation(name, lanes):
    spec, m, gold = spec_for(name)
    s = A.Sampler(spec, chains=96, seed=gold["case"]["seed"], lanes_per_chain=lanes)
    assert s.launch_info()["lanes_per_chain"] == lanes
    s.burn(120)
    draws = s.sample(30, 3)
    st, lp = s.state(), s.diag()["log_post"]
    # (the summation order the sampler's values follow: its lane count -- or 1, the reference's own, for the kernels that decide against the expression in that order:
    # a closure with a certified row plan at 64 lanes, a certified tail at one)
    order = s.launch_info()["summation_order"]
    assert order in (1, lanes)
    for c in range(0, 96, 5):
        want = m.eval(st[:, c], order)
        assert np.float64(lp[c]).tobytes() == np.float64(want).tobytes(), (name, lanes, c, lp[c], want)
    if m.meta["derived"]:
        for t in (0, 9):
            for c in (0, 50, 95):
                _, dv = m.eval(draws[t, : s.P, c], 1, derived=True)
                assert draws[t, s.P:, c].tolist() == dv
    assert np.all(np.isfinite(lp))
    if lanes == 1:      # decisions of chain 0 are the reference's
        rec = gold["chains"][0]
        if rec["chain"] == 0:
            s2 = A.Sampler(spec, chains=2, seed=gold["case"]["seed"], lanes_per_chain=1)
            n = sum(seg["n"] for seg in gold["case"]["schedule"])
            s2.burn(n)
            assert s2.info()["accepts"][:, 0].tolist() == rec["accepts"]
            assert s2.state()[:, 0].tolist() == rec["final_state"]
            s2.close()
    s.close()


@pytest.mark.parametrize("name,lanes", [("complex_model", 4), ("spike_slab", 8), ("hier_binomial", 2), ("norm_post_derived", 16), ("survival_mix", 4),
                                        ("complex_model", 128), ("norm_post_derived", 256), ("spike_slab", 256),
                                        ("structured_helpers", 8), ("records_logistic", 2), ("modern_js", 1)])
def test_g_lane_trajectories_equal_oracle_stepper_with_same_lane_order(name, lanes):
    """Whole trajectories with G lanes per chain: the device == the C oracle's stepper calling the host build of the same
    generated text in the same lane order (test_translate.py pins that pair against the reference at one lane)."""
    import oracle_lib
    from gpu_util import assert_chain_equals_oracle, run_schedule
    from test_translate import oracle_spec
    spec, m, gold = spec_for(name)
    ospec, _, _ = oracle_spec(name)
    seed = 1234
    s = A.Sampler(spec, chains=40, seed=seed, chain_offset=7, lanes_per_chain=lanes)
    sched = [{"op": "burn", "n": 110}, {"op": "sample", "n": 45, "thin": 2}]
    gs = run_schedule(s, sched)
    P = s.P
    for local in (0, 39):
        o = oracle_lib.OracleChain(ospec, seed, 7 + local, lanes=lanes)
        os_ = run_schedule(o, sched)
        assert_chain_equals_oracle(s, local, o, [g[:, :P, :] for g in gs], os_)
    s.close()


@pytest.mark.parametrize("name,builtin", [("hier_normal_closure", "hier_small"), ("pois_glm_closure", "glm_small")])
@pytest.mark.parametrize("lanes", [1, 4, 16, 64])
def test_translated_closure_equals_hand_written_family_at_any_lane_count(name, builtin, lanes):
    """Two independent implementations of the same model -- the hand-written functor of csrc/amwg_models.h and the text
    translate.js generates from the closure -- use the same lane order and give the same bits."""
    import model_spec
    from gpu_util import run_schedule
    spec, m, gold = spec_for(name)
    bgold = golden_io.load(builtin)
    bspec = model_spec.spec_from_golden(bgold, bgold["chains"][0])
    sched = [{"op": "burn", "n": 90}, {"op": "sample", "n": 40, "thin": 2}]
    kw = dict(chains=24, seed=4321, chain_offset=3, lanes_per_chain=lanes)
    a, b = A.Sampler(spec, **kw), A.Sampler(bspec, **kw)
    da, db = run_schedule(a, sched), run_schedule(b, sched)
    assert da[0].tobytes() == db[0].tobytes()
    assert a.state().tobytes() == b.state().tobytes()
    assert a.info()["accepts"].tolist() == b.info()["accepts"].tolist()
    oa, ob = a.launch_info()["summation_order"], b.launch_info()["summation_order"]
    if oa != ob:
        # the hand-written family's default at this lane count decides against the expression in the REFERENCE's order (certified kernels, round 5) and the
        # closure's does not: same draws, log_post a last-bits neighbour; the kernel that sums in the closure's lane order is options.full_evaluation = 1
        assert ob == 1 and lanes > 1 and oa == lanes
        b.close()
        b = A.Sampler(bspec, full_evaluation=1, **kw)
        db = run_schedule(b, sched)
        assert da[0].tobytes() == db[0].tobytes() and a.state().tobytes() == b.state().tobytes()
    # (round 6: a closure with a certified row plan decides in the reference's order too -- both sides then hold the reference's own log_post)
    assert a.diag()["log_post"].tobytes() == b.diag()["log_post"].tobytes()
    a.close(); b.close()


def _golden_spec(gold, rec, src, arrays, meta):
    params, init = [], []
    for p in rec["params_completed"]:
        ln = int(np.prod(p["dim"]))
        params.append({"type": p["type"], "len": ln, "top": p["dim"][0], "multidim": 0 if p["dim"] == [1] else 1, "lower": p["lower"], "upper": p["upper"]})
        init += p["init"]
    return {"user": user_host.user_spec_part(src, arrays, meta), "params": params, "P": len(init), "init": init, "comp_opts": rec["comp_opts"]}


@pytest.mark.parametrize("name,kernel", [("hier_normal_closure", "amwg_user_sweep_cert"), ("hier_rows_bounded", "amwg_user_sweep_cert"), ("hier_rows_int", "amwg_user_sweep_cert")])
def test_row_plan_of_a_translated_closure_reproduces_the_reference_and_the_full_evaluation(name, kernel):
    """csrc/amwg_rows.h on the device, 64 lanes per chain: a closure that ends in the likelihood loop of a model with group means keeps the per-lane sums an
    update cannot have changed and evaluates the proposals of a whole sweep over theta in one pass -- for a swept vector that is not the first parameter, has
    bounds (a proposal outside draws no accept uniform) or is of integer type as well.  Checked against (1) the seeded run of the UNMODIFIED reference
    (tests/golden/user_<name>.json): accept counts, in-bounds counts, adaptation state and uniforms consumed of the golden's chains; (2) the same sampler with
    options.full_evaluation = 1 (every update evaluates the whole closure): every draw, the final state and the cached log_post bit for bit, 96 chains."""
    from gpu_util import run_schedule
    gold = golden_io.load("user_" + name)
    m = user_host.host_model(name)
    assert m.meta["rows_n_obs"] > 0
    for rec in gold["chains"]:
        spec = _golden_spec(gold, rec, m.source, m.arrays, m.meta)
        s = A.Sampler(spec, chains=3, seed=gold["case"]["seed"], chain_offset=rec["chain"], lanes_per_chain=64)
        assert s.launch_info()["kernel"] == kernel and s.launch_info()["lanes_per_chain"] == 64
        run_schedule(s, gold["case"]["schedule"])
        info = s.info()
        assert info["accepts"][:, 0].tolist() == rec["accepts"] and info["inbounds"][:, 0].tolist() == rec["inbounds"]
        assert info["batch_count"][:, 0].tolist() == rec["batch_count"] and int(s.diag()["uniforms"][0]) == rec["uniforms"]
        # (round 6: the row plan's certified kernel decides against the expression in the REFERENCE's order and leaves that expression's value behind: the golden's
        # own doubles, not merely its decisions)
        assert s.launch_info()["summation_order"] == 1
        assert s.state()[:, 0].tolist() == rec["final_state"] and float(s.diag()["log_post"][0]) == rec["log_post"]
        s.close()
    spec = _golden_spec(gold, gold["chains"][0], m.source, m.arrays, m.meta)
    sched = [{"op": "burn", "n": 130}, {"op": "sample", "n": 60, "thin": 2}, {"op": "burn", "n": 7}]
    kw = dict(chains=96, seed=77, chain_offset=5, lanes_per_chain=64)
    # (full_evaluation = 2: the row plan with every sweep's accept tests decided update by update -- the path a sweep takes when a uniform falls inside the
    # rounding bound of the all-at-once decision, csrc/amwg_kernel.h)
    # (c3 / c4: the certified bounds widened 2^12- and 2^40-fold -- fallbacks to the expression in the reference's order often / always; r1: ONE lane per chain with the
    # expression in every update = the reference's own order: the certified kernel's log_post must be ITS value, bit for bit)
    a, b, c2, c3, c4 = A.Sampler(spec, **kw), A.Sampler(spec, full_evaluation=1, **kw), A.Sampler(spec, full_evaluation=2, **kw), A.Sampler(spec, test_bound_shift=12, **kw), A.Sampler(spec, test_bound_shift=40, **kw)
    r1 = A.Sampler(spec, full_evaluation=1, **dict(kw, lanes_per_chain=1))
    assert a.launch_info()["kernel"] == kernel and b.launch_info()["kernel"] == "amwg_user_step" and c2.launch_info()["kernel"] == "amwg_user_sweep" and c3.launch_info()["kernel"] == kernel
    da = run_schedule(a, sched)
    for o in (b, c2, c3, c4, r1):
        do = run_schedule(o, sched)
        assert da[0].tobytes() == do[0].tobytes()
        assert a.state().tobytes() == o.state().tobytes()
        if o in (c3, c4, r1):
            assert a.diag()["log_post"].tobytes() == o.diag()["log_post"].tobytes()
        else:      # the lane-order kernels: the same chains, log_post in the last bits apart
            assert np.allclose(a.diag()["log_post"], o.diag()["log_post"], rtol=1e-11, atol=0)
        assert a.info()["accepts"].tolist() == o.info()["accepts"].tolist() and a.diag()["uniforms"].tolist() == o.diag()["uniforms"].tolist()
    for q in (a, b, c2, c3, c4, r1):
        q.close()


def test_translated_hierarchical_closure_at_full_size_runs_the_sweep_kernel_and_equals_the_hand_written_family():
    """BASELINE.json configs[3] written as a plain closure, 64 lanes per chain, 2 048 chains: the translated closure runs amwg_user_sweep (row plan + sweep
    prefetch) and gives the same bits as the hand-written family's lane-order sweep kernel -- two implementations of the same 64-lane order --, the draws of its
    certified sweep kernel, and the decisions of the seeded reference run (cfg4_full: chain ids 0 and 16383)."""
    import model_spec
    from gpu_util import run_schedule
    gold = golden_io.load("cfg4_full")
    src, arrays, meta = user_host.translated("bench_hier")
    assert (meta["rows_n_obs"], meta["rows_groups"], meta["rows_sweep"], meta["rows_cert"]) == (10000, 32, 1, 1)
    for rec in gold["chains"]:
        spec = _golden_spec(gold, rec, src, arrays, meta)
        bspec = model_spec.spec_from_golden(gold, rec)
        kw = dict(chains=4, seed=gold["case"]["seed"], chain_offset=rec["chain"], lanes_per_chain=64)
        # (b: the hand-written family's sweep kernel in the same 64-lane order -- options.full_evaluation = 2; c: its default, the certified sweep kernel, which
        # decides against the expression in the reference's order: same draws, log_post in the last bits apart)
        # (a: the closure's lane-order sweep kernel -- options.full_evaluation = 2; b: the hand-written family's, same 64-lane order; c: the family's default, the certified
        # sweep kernel; e: round 6, the CLOSURE's default -- the translator's certified row plan, amwg_user_sweep_cert: decides against the expression in the reference's
        # order like c: same draws as all of them, log_post the reference's own)
        a, b, c, e = A.Sampler(spec, full_evaluation=2, **kw), A.Sampler(bspec, full_evaluation=2, **kw), A.Sampler(bspec, **kw), A.Sampler(spec, **kw)
        assert a.launch_info()["kernel"] == "amwg_user_sweep" and b.launch_info()["kernel"].startswith("amwg_sweep_kernel<") and c.launch_info()["kernel"].startswith("amwg_sweep_kernel_cert<")
        assert e.launch_info()["kernel"] == "amwg_user_sweep_cert" and e.launch_info()["summation_order"] == 1
        da, db, dc, de = (run_schedule(q, gold["case"]["schedule"]) for q in (a, b, c, e))
        assert all(x.tobytes() == y.tobytes() == z.tobytes() == w.tobytes() for x, y, z, w in zip(da, db, dc, de))
        assert a.state().tobytes() == b.state().tobytes() == c.state().tobytes() == e.state().tobytes() and a.diag()["log_post"].tobytes() == b.diag()["log_post"].tobytes()
        assert float(c.diag()["log_post"][0]) == rec["log_post"] and e.diag()["log_post"].tobytes() == c.diag()["log_post"].tobytes()      # the reference's own value
        info = e.info()
        assert info["accepts"][:, 0].tolist() == rec["accepts"] and info["inbounds"][:, 0].tolist() == rec["inbounds"] and int(e.diag()["uniforms"][0]) == rec["uniforms"]
        a.close(); b.close(); c.close(); e.close()


def test_certified_tail_of_a_translated_closure_decides_like_the_expression_and_the_family():
    """Round 6: a closure that ENDS in `for (i) lp += ld.norm(x[i], mean, sd)` over an f64 array gets certified decisions from the translator (translate.js tailPlan,
    csrc/amwg_user.h norm_tail_approx): with one lane per chain amwg_user_step_cert decides from head + n c - S2 / den and its bound, and evaluates the closure
    itself where that does not decide.  BASELINE.json configs[1] written as a plain closure (not recognised as the family: translated): the default equals the
    expression in every update (full_evaluation = 1: amwg_user_step), the bounds widened 2^14- and 2^40-fold (fallbacks often / always), the hand-written family
    on the same data, and the seeded reference run (cfg2_full) -- draws, state, counters, uniforms, cached log_post, bit for bit (mcmc.js:524-528)."""
    import model_spec
    gold = golden_io.load("cfg2_full")
    src, arrays, meta = user_host.translated("bench_normal")
    assert meta["cert_tail_n"] == 10000 and "kCertifiedTail = true" in src
    for rec in gold["chains"]:
        spec, bspec = _golden_spec(gold, rec, src, arrays, meta), model_spec.spec_from_golden(gold, rec)
        kw = dict(chains=320, seed=gold["case"]["seed"], chain_offset=rec["chain"], lanes_per_chain=1)
        runs = [A.Sampler(spec, **kw), A.Sampler(spec, full_evaluation=1, **kw), A.Sampler(spec, test_bound_shift=14, **kw), A.Sampler(spec, test_bound_shift=40, **kw), A.Sampler(bspec, **kw)]
        names = [q.launch_info()["kernel"] for q in runs]
        assert names[0] == names[2] == names[3] == "amwg_user_step_cert" and names[1] == "amwg_user_step" and names[4].startswith("amwg_step_kernel_cert<NormalModel,1,"), names
        assert all(q.launch_info()["summation_order"] == 1 for q in runs)
        outs = []
        for q in runs:
            q.burn(57)
            d1 = q.sample(40, 1)
            q.burn(130)
            d2 = q.sample(21, 3)
            outs.append((d1.tobytes(), d2.tobytes(), q.state().tobytes(), q.info()["accepts"].tobytes(), q.info()["prop_log_scale"].tobytes(), q.diag()["uniforms"].tobytes(), q.diag()["log_post"].tobytes()))
        assert all(o == outs[0] for o in outs[1:]), [[x == y for x, y in zip(o, outs[0])] for o in outs[1:]]
        for q in runs:
            q.close()
    # ... and out of the golden's own schedule: the reference's chain, bit for bit
    from gpu_util import run_schedule
    rec = gold["chains"][0]
    s = A.Sampler(_golden_spec(gold, rec, src, arrays, meta), chains=64, seed=gold["case"]["seed"], chain_offset=rec["chain"], lanes_per_chain=1)
    assert s.launch_info()["kernel"] == "amwg_user_step_cert"
    for got, want in zip(run_schedule(s, gold["case"]["schedule"]), rec["samples"]):
        w = np.array(want["draws"], dtype=np.float64)
        assert np.ascontiguousarray(got[: w.shape[0], :, 0]).tobytes() == w.tobytes()
    assert s.state()[:, 0].tolist() == rec["final_state"] and s.info()["accepts"][:, 0].tolist() == rec["accepts"] and float(s.diag()["log_post"][0]) == rec["log_post"]
    s.close()


@pytest.mark.parametrize("name,params,init,n", [
    ("bench_normal_50k", [("real", -INF, INF), ("real", 0.0, INF)], [0.5, 0.5], 50000),      # 400 KB of observations: the wavefront's pass reads global memory
    ("bench_normal_n65", [("real", -INF, INF), ("real", 0.0, INF)], [0.5, 0.5], 65),         # one full round and one observation
    ("bench_normal_expr", [("real", -INF, INF), ("real", -INF, INF), ("real", 0.0, INF)], [0.5, 0.5, 1.0], 3000)])      # mean = a + b / 2, sd = 1 / sqrt(tau), a gamma prior in the head
def test_certified_tail_beyond_the_readme_shape(name, params, init, n):
    """translate.js tailPlan on closures that are not the README's: the default (amwg_user_step_cert) against the expression in every update and widened bounds --
    draws, state, counters, proposal scales, uniforms, cached log_post: every bit of every chain."""
    src, arrays, meta = user_host.translated(name)
    assert meta["cert_tail_n"] == n and "kCertifiedTail = true" in src
    opt = {"prop_log_scale": 0.0, "batch_size": 50, "max_adaptation": 0.33, "initial_adaptation": 1.0, "target_accept_rate": 0.44, "is_adapting": True}
    spec = {"user": user_host.user_spec_part(src, arrays, meta), "P": len(init), "init": init, "comp_opts": [dict(opt) for _ in init],
            "params": [{"type": t, "len": 1, "top": 1, "multidim": 0, "lower": lo, "upper": hi} for t, lo, hi in params]}
    chains = 192 if n > 10000 else 640
    kw = dict(chains=chains, seed=5, chain_offset=11, lanes_per_chain=1, steps_per_launch=17)
    runs = [A.Sampler(spec, **kw), A.Sampler(spec, full_evaluation=1, **kw), A.Sampler(spec, test_bound_shift=16, **kw), A.Sampler(spec, test_bound_shift=40, **kw)]
    assert [q.launch_info()["kernel"] for q in runs] == ["amwg_user_step_cert", "amwg_user_step", "amwg_user_step_cert", "amwg_user_step_cert"]
    outs = []
    for q in runs:
        d1 = q.sample(60, 2)
        q.burn(90)
        d2 = q.sample(30, 1)
        outs.append((d1.tobytes(), d2.tobytes(), q.state().tobytes(), q.info()["accepts"].tobytes(), q.info()["prop_log_scale"].tobytes(), q.diag()["uniforms"].tobytes(), q.diag()["log_post"].tobytes()))
        q.close()
    assert all(o == outs[0] for o in outs[1:]), [[x == y for x, y in zip(o, outs[0])] for o in outs[1:]]
    assert np.isfinite(np.frombuffer(outs[0][2], dtype=np.float64)).all()


def test_certified_poisson_tail_of_a_translated_closure_reproduces_the_reference_and_the_full_evaluation():
    """translate.js poisTailPlan + csrc/amwg_ptail.h: a closure that ends in `lp += ld.pois(y[i], Math.exp(eta))` runs amwg_user_step_cert at 16 lanes per chain (four
    chains of a wavefront share every row).  Its decisions are the expression's in the REFERENCE's order: chain by chain the reference's golden trajectory, and every bit
    of the run that evaluates the expression in every update at one lane per chain; widened and narrowed bounds change nothing."""
    from gpu_util import run_schedule
    spec, m, gold = spec_for("pois_glm_closure")
    assert m.meta["pois_tail_n"] == 500 and "kPoisTail = true" in m.source and "kTailUniformState = true" in m.source
    sched = gold["case"]["schedule"]
    seed = gold["case"]["seed"]
    auto = A.Sampler(spec, chains=8192, seed=seed)      # (enough chains for the work model to look at lane counts at all)
    assert (auto.launch_info()["lanes_per_chain"], auto.launch_info()["kernel"], auto.launch_info()["summation_order"]) == (16, "amwg_user_step_cert", 1)
    auto.close()
    kw = dict(chains=64, seed=seed, steps_per_launch=23)
    runs = [A.Sampler(spec, lanes_per_chain=16, **kw), A.Sampler(spec, lanes_per_chain=1, full_evaluation=1, **kw),
            A.Sampler(spec, lanes_per_chain=16, test_bound_shift=12, **kw), A.Sampler(spec, lanes_per_chain=16, test_bound_shift=30, **kw)]
    assert [q.launch_info()["kernel"] for q in runs] == ["amwg_user_step_cert", "amwg_user_step", "amwg_user_step_cert", "amwg_user_step_cert"]
    outs = []
    for q in runs:
        segs = run_schedule(q, sched)
        outs.append((b"".join(g.tobytes() for g in segs), q.state().tobytes(), q.info()["accepts"].tobytes(), q.info()["prop_log_scale"].tobytes(), q.diag()["uniforms"].tobytes(), q.diag()["log_post"].tobytes()))
    assert all(o == outs[0] for o in outs[1:]), [[x == y for x, y in zip(o, outs[0])] for o in outs[1:]]
    for rec in gold["chains"]:      # the reference's own chains
        c = rec["chain"]
        assert runs[0].info()["accepts"][:, c].tolist() == rec["accepts"]
        assert runs[0].state()[:, c].tolist() == rec["final_state"]
        assert float(runs[0].diag()["log_post"][c]) == rec["log_post"]      # (the expression in the reference's order: the reference's own double)
    for q in runs:
        q.close()


@pytest.mark.parametrize("name,flags", [("pois_tail_nonlinear", ("true", "true", "false")), ("pois_tail_gather", ("false", "false", "false")), ("pois_tail_next_row", ("true", "false", "false"))])
def test_certified_poisson_tail_fallback_paths_equal_the_expression(name, flags):
    """csrc/amwg_ptail.h off its fast path -- a predictor that is not linear (eta by the closure's own statements, H = max |eta| over the rows), a coefficient gathered
    by the data (per-lane LDS reads of the state, the plain loop), a read of the next observation's row (no row cache) --, 517 observations (a ragged last round): the
    16-lane certified kernel against the same closure at ONE lane per chain with the expression in every update, and against narrowed / widened bounds: every bit of
    every chain, cached log_post included (the certified kernel evaluates the expression in the reference's order)."""
    import re
    src, arrays, meta = user_host.translated(name)
    assert meta["pois_tail_n"] == 517
    got = tuple(re.search(k + r" = (true|false)", src).group(1) for k in ("kTailUniformState", "kTailRows", "kTailLinear"))
    assert got == flags, got
    opt = {"prop_log_scale": 0.0, "batch_size": 50, "max_adaptation": 0.33, "initial_adaptation": 1.0, "target_accept_rate": 0.44, "is_adapting": True}
    params = [{"type": "real", "len": 8, "top": 8, "multidim": 1, "lower": -INF, "upper": INF}, {"type": "int", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": 516.0}]
    spec = {"user": user_host.user_spec_part(src, arrays, meta), "params": params, "P": 9, "init": [0.1] * 8 + [250.0], "comp_opts": [dict(opt) for _ in range(9)]}
    kw = dict(chains=96, seed=11, chain_offset=5, steps_per_launch=19)
    runs = [A.Sampler(spec, lanes_per_chain=16, **kw), A.Sampler(spec, lanes_per_chain=1, full_evaluation=1, **kw),
            A.Sampler(spec, lanes_per_chain=16, test_bound_shift=10, **kw), A.Sampler(spec, lanes_per_chain=16, test_bound_shift=34, **kw)]
    assert [q.launch_info()["kernel"] for q in runs] == ["amwg_user_step_cert", "amwg_user_step", "amwg_user_step_cert", "amwg_user_step_cert"]
    assert runs[0].launch_info()["summation_order"] == 1
    outs = []
    for q in runs:
        d1 = q.sample(70, 2)
        q.burn(110)
        d2 = q.sample(25, 1)
        outs.append((d1.tobytes(), d2.tobytes(), q.state().tobytes(), q.info()["accepts"].tobytes(), q.info()["prop_log_scale"].tobytes(), q.diag()["uniforms"].tobytes(), q.diag()["log_post"].tobytes()))
        q.close()
    assert all(o == outs[0] for o in outs[1:]), [[x == y for x, y in zip(o, outs[0])] for o in outs[1:]]
    assert np.isfinite(np.frombuffer(outs[0][2], dtype=np.float64)).all()


FULL_SIZE = [("bench_normal", "cfg2_full"), ("bench_bern", "cfg3_full"), ("bench_hier", "cfg4_full"), ("bench_glm", "cfg5_full")]


@pytest.fixture(scope="module")
def full_size_runs(request):
    """the selected full-size closures, one lane per chain, every chain of every closure side by side (a one-lane chain at N = 5e4 is one wavefront for
    most of a minute: run one after the other these were 60 s of the suite) -> {closure: [(chain record, sampler, draw segments)]}"""
    from gpu_util import run_schedules_concurrently
    want = {it.callspec.params["closure"] for it in request.session.items
            if getattr(it, "originalname", "") == "test_translated_closures_reproduce_the_reference_at_full_baseline_sizes" and hasattr(it, "callspec")}
    jobs, meta = [], []
    for closure, golden in FULL_SIZE:
        if closure not in want:
            continue
        gold = golden_io.load(golden)
        src, arrays, m = user_host.translated(closure)
        # (cfg5: the LAST chain id only -- the built-in family's test walks both ids)
        for rec in (gold["chains"][-1:] if golden == "cfg5_full" else gold["chains"]):
            s = A.Sampler(_golden_spec(gold, rec, src, arrays, m), chains=2, seed=gold["case"]["seed"], chain_offset=rec["chain"], lanes_per_chain=1)
            jobs.append((s, gold["case"]["schedule"]))
            meta.append((closure, rec, s))
    segs = run_schedules_concurrently(jobs)
    out = {}
    for (closure, rec, s), sg in zip(meta, segs):
        out.setdefault(closure, []).append((rec, s, sg))
    yield out
    for closure, rec, s in meta:
        s.close()


@pytest.mark.parametrize("closure,golden", FULL_SIZE)
def test_translated_closures_reproduce_the_reference_at_full_baseline_sizes(closure, golden, full_size_runs):
    """BASELINE.json configs[1..4] at their full data sizes, written as plain closures and TRANSLATED: one lane per chain
    reproduces the seeded runs of the unmodified reference (first and last chain id of each config) bit for bit."""
    for rec, s, segs in full_size_runs[closure]:
        for got, want in zip(segs, rec["samples"]):
            w = np.array(want["draws"], dtype=np.float64)
            assert np.ascontiguousarray(got[: w.shape[0], :, 0]).tobytes() == w.tobytes()
            tot = np.zeros(got.shape[1])
            for t in range(got.shape[0]):
                tot = tot + got[t, :, 0]
            assert tot.tolist() == want["sum"]
        assert s.state()[:, 0].tolist() == rec["final_state"]
        assert s.info()["accepts"][:, 0].tolist() == rec["accepts"]
        assert int(s.diag()["uniforms"][0]) == rec["uniforms"]
        assert float(s.diag()["log_post"][0]) == rec["log_post"]


@pytest.mark.parametrize("name,chains,steps,lanes", [("logit_n10k", 512, 500, 16), ("logit_bern_n10k", 256, 400, 64), ("logistic_softplus", 8192, 10_000, 4)])
def test_translated_logistic_decisions_at_many_lanes_equal_the_one_lane_run(name, chains, steps, lanes):
    """Translated closures, decision parity counted as for the built-in families (tests/decision_parity.py): the same seeded job with one lane per
    chain (the reference's summation order) and lane-split -- where the fused softplus runs in its branch-free form, four terms per flag test, several
    chains to a wavefront -- chain against chain on the device; at most 50 first flips per 1e9 decisions."""
    import math
    import decision_parity as dp
    spec, m, gold = spec_for(name)
    r = dp.compare(A, spec, chains, steps, seed=gold["case"]["seed"], alt={"lanes_per_chain": lanes})
    assert r["reference_geometry"]["lanes_per_chain"] == 1 and r["geometry"]["lanes_per_chain"] == lanes
    assert r["decisions"] > 0.8 * chains * steps * r["components"]
    assert r["chains_differing"] <= math.ceil(50.0 * 1e-9 * r["decisions"]), r
    assert r["lp_abs_diff_max"] <= 1e-10 * max(1.0, r["lp_abs_typical"]), r


def test_translated_normal_samples_the_analytic_posterior():
    """Normal model with flat-ish priors: posterior mean of mu ~ data mean, E[sigma^2] ~ s^2 (n-1)/(n-3)."""
    spec, m, gold = spec_for("norm_post_derived")
    x = np.array(m.arrays[0])
    s = A.Sampler(spec, chains=4096, seed=99, lanes_per_chain=2)
    s.burn(1500)
    d = s.sample(400, 4)
    mu, var = d[:, 0, :].ravel(), d[:, 2, :].ravel()
    n = x.size
    # mu ~ norm(0,100) pulls the mean by ~1 %, unif(0,100) truncates the sd's tail: generous but non-trivial bounds
    assert abs(mu.mean() - x.mean()) < 0.08 * x.std()
    assert abs(var.mean() / (x.var(ddof=1) * (n - 1) / (n - 4)) - 1) < 0.12   # E[sigma^2 | x] under p(sigma) uniform
    s.close()


@pytest.mark.parametrize("lanes", [1, 4])
def test_fuzzed_closures_on_the_device_equal_v8(lanes):
    """Random closures (tests/js/fuzz_translate_cli.js, see test_translate.py) compiled with hiprtc and evaluated by the step kernel itself:
    40 chains start at the 40 random states; the cached log_post (the constructor's warm-up evaluation) and the derived quantities of
    the first recorded draw are, at one lane per chain, bit for bit what V8 returned, and at four lanes bit for bit what the host
    build of the same text returns in that lane order."""
    # 12 derived quantities per model, seeds whose programs hiprtc compiles in seconds (the 48-quantity models of the host test take a minute each)
    # (AMWG_FUZZ_SEEDS="30,31,..." runs a one-off campaign over other seeds)
    seeds = [int(t) for t in os.environ.get("AMWG_FUZZ_SEEDS", "5,7,25").split(",")]
    for name in [nm for sd in seeds for nm in user_host.fuzz_models(sd, 1, 12)]:
        m = user_host.host_model(name)
        if lanes > 1 and not m.meta["parallel"]:
            continue
        pts = user_host.stepper_states(name)
        f64 = lambda h: float(np.frombuffer(bytes.fromhex(h), dtype=">f8")[0])
        states = np.array([[f64(h) for h in pt["state"]] for pt in pts])            # [40][13]: a, b, v[3], k, z, w[2][3]
        params = [{"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": -INF, "upper": INF}, {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": INF},
                  {"type": "real", "len": 3, "top": 3, "multidim": 1, "lower": -INF, "upper": INF}, {"type": "int", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": 6.0},
                  {"type": "binary", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": 1.0},
                  {"type": "real", "len": 6, "top": 2, "multidim": 1, "lower": -INF, "upper": INF}]
        NP = 13
        opts = [{"prop_log_scale": 0.0, "max_adaptation": 0.33, "initial_adaptation": 1.0, "target_accept_rate": 0.44, "batch_size": 50, "is_adapting": True}] * NP
        spec = {"user": user_host.user_spec_part(m.source, m.arrays, m.meta), "params": params, "P": NP, "init": states[0].tolist(), "comp_opts": opts}
        s = A.Sampler(spec, chains=len(pts), seed=1, lanes_per_chain=lanes)
        s.set_state(np.ascontiguousarray(states.T))
        lp = s.diag()["log_post"]
        draws = s.sample(1, 1)
        assert draws[0, :NP, :].tobytes() == np.ascontiguousarray(states.T).tobytes()
        for c, pt in enumerate(pts):
            if lanes == 1:
                want_lp, want_dv = f64(pt["lp"]), [f64(h) for h in pt["derived"]]
            else:
                want_lp, want_dv = m.eval(states[c], lanes, derived=True)
            same = lambda a, b: (a != a and b != b) or np.float64(a).tobytes() == np.float64(b).tobytes()
            assert same(lp[c], want_lp), (name, lanes, c, lp[c], want_lp)
            got_dv = draws[0, NP:, c].tolist()
            assert len(got_dv) == len(want_dv) and all(same(a, b) for a, b in zip(got_dv, want_dv)), (name, lanes, c)
        s.close()


def test_autotuned_translated_closure_equals_the_plain_construction():
    """AMWG_LANES_AUTOTUNE on a translated closure: one hiprtc compile per candidate lane count, timing runs on the real chain state
    (saved and restored), the module of the winner kept -- the sampler then behaves exactly like one constructed with that lane count."""
    spec, m, gold = spec_for("hier_normal_closure")
    tuned = A.Sampler(spec, chains=256, seed=9, lanes_per_chain=-2)
    cands = tuned.tuning()
    lanes = tuned.launch_info()["lanes_per_chain"]
    assert len(cands) >= 3 and lanes in [c[0] for c in cands]
    plain = A.Sampler(spec, chains=256, seed=9, lanes_per_chain=lanes)
    for s_ in (tuned, plain):
        s_.burn(80)
    a, b = tuned.sample(30, 3), plain.sample(30, 3)
    assert a.tobytes() == b.tobytes()
    assert tuned.info()["accepts"].tobytes() == plain.info()["accepts"].tobytes()
    tuned.close()
    plain.close()
