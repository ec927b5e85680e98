"""-m gpu: the HIP path, through the C ABI, against the reference goldens and the oracle.

Bars (north_star): integer/decision state bit-identical; with one lane per chain (the
reference's summation order) every double bit-identical to the REFERENCE; with G lanes per
chain every double bit-identical to the oracle run in the same G-lane order, and every accept
decision still identical to the reference's.
"""
import os

import numpy as np
import pytest

import amwg_ctypes as A
import golden_io
import model_spec
import oracle_lib
from gpu_util import assert_chain_equals_oracle, run_schedule, run_schedule_many

pytestmark = pytest.mark.gpu

GOLDEN_CASES = ["cfg1_heights", "normal_n1000", "cfg2_full", "normal_opts", "beta_bern_n2000", "cfg3_full", "hier_small",
                "cfg4_full", "glm_small", "cfg5_full", "normal_hyper", "beta_bern_hyper", "beta_bern_hyper2", "hier_hyper", "glm_hyper",
         "cfg4_theta_bounded", "cfg4_theta_int"]      # (configs[3] with a bounded / an integer theta: updates that draw no accept uniform, rounded proposals)


@pytest.fixture(scope="module")
def one_lane_runs(request):
    """Every selected golden case's chains with ONE lane per chain, all run SIDE BY SIDE (one host thread and one stream per sampler): a one-lane chain of
    cfg5 is a single wavefront for over a minute, and run one case after the other these tests were a third of the suite's time (round-4 review).
    -> {case: [(chain record, sampler, draw segments)]}"""
    from gpu_util import run_schedules_concurrently
    names = sorted({it.callspec.params["name"] for it in request.session.items
                    if getattr(it, "originalname", "") == "test_one_lane_per_chain_is_bit_identical_to_reference" and hasattr(it, "callspec")})
    jobs, meta = [], []
    for name in names:
        gold = golden_io.load(name)
        case = gold["case"]
        for rec in gold["chains"]:
            s = A.Sampler(model_spec.spec_from_golden(gold, rec), chains=3, seed=case["seed"], chain_offset=rec["chain"], lanes_per_chain=1)
            jobs.append((s, case["schedule"]))
            meta.append((name, rec, s))
    segs = run_schedules_concurrently(jobs)
    out = {}
    for (name, rec, s), sg in zip(meta, segs):
        out.setdefault(name, []).append((rec, s, sg))
    yield out
    for name, rec, s in meta:
        s.close()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_one_lane_per_chain_is_bit_identical_to_reference(name, one_lane_runs):
    for rec, s, segs in one_lane_runs[name]:
        for got, want in zip(segs, rec["samples"]):
            assert got.shape[0] == want["kept"]
            w = np.array(want["draws"], dtype=np.float64).reshape(-1, got.shape[1])
            assert np.ascontiguousarray(got[: w.shape[0], :, 0]).tobytes() == w.tobytes()
            tot = np.zeros(got.shape[1])
            for t in range(got.shape[0]):
                tot = tot + got[t, :, 0]
            assert tot.tolist() == want["sum"]
        info, d = s.info(), s.diag()
        assert s.state()[:, 0].tolist() == rec["final_state"]
        assert info["accepts"][:, 0].tolist() == rec["accepts"]
        assert info["inbounds"][:, 0].tolist() == rec["inbounds"]
        assert info["prop_log_scale"][:, 0].tolist() == rec["prop_log_scale"]
        assert info["batch_count"][:, 0].tolist() == rec["batch_count"]
        assert info["acceptance_count"][:, 0].tolist() == rec["acceptance_count"]
        assert info["iterations_since_adaption"][:, 0].tolist() == rec["iterations_since_adaption"]
        assert int(d["uniforms"][0]) == rec["uniforms"]
        assert d["named_order"][0].tolist() == rec["named_order"]
        assert float(d["log_post"][0]) == rec["log_post"]


@pytest.mark.parametrize("name,lanes", [(n, l) for n in ["cfg1_heights", "normal_n1000", "normal_opts", "beta_bern_n2000", "hier_small", "glm_small"]
                                        for l in [2, 4, 8, 16, 32, 64]] +
                         [("cfg4_full", 64), ("cfg4_full", 32), ("cfg5_full", 64), ("cfg5_full", 16),      # the full-size cases at the lane counts the bench times
                          ("cfg4_theta_bounded", 64), ("cfg4_theta_int", 64)])           # ... the sweep kernel straight against reference goldens with a bounded / integer theta
def test_g_lanes_per_chain_matches_oracle_and_reference_decisions(name, lanes):
    gold = golden_io.load(name)
    case = gold["case"]
    rec = gold["chains"][0]
    spec = model_spec.spec_from_golden(gold, rec)
    s = A.Sampler(spec, chains=5, seed=case["seed"], chain_offset=rec["chain"], lanes_per_chain=lanes)
    # (the oracle sums in the order the sampler does: `lanes` partial sums and a butterfly -- or, for the kernels that decide against the expression in the
    # reference's own order (the hierarchical sweep kernel at 64 lanes, the Poisson family at 16), ONE running sum: the cached log_post is then the reference's too)
    order = s.launch_info()["summation_order"]
    assert order == (1 if (spec["model"], lanes) == ("pois_glm", 16) or s.launch_info()["kernel"].startswith("amwg_sweep_kernel") else lanes)
    o = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=order)
    gs, os_ = run_schedule(s, case["schedule"]), run_schedule(o, case["schedule"])
    assert_chain_equals_oracle(s, 0, o, gs, os_)
    if order == 1:
        assert float(s.diag()["log_post"][0]) == rec["log_post"]
    assert s.info()["accepts"][:, 0].tolist() == rec["accepts"]      # the reference's decisions
    assert s.info()["inbounds"][:, 0].tolist() == rec["inbounds"]
    assert int(s.diag()["uniforms"][0]) == rec["uniforms"]
    if name.startswith("cfg4_theta"):
        assert s.launch_info()["kernel"].startswith("amwg_sweep_kernel")
    s.close()


@pytest.mark.parametrize("name", ["normal_n1000", "beta_bern_n2000", "hier_small", "glm_small"])
@pytest.mark.parametrize("lanes", [128, 256, 1024])
def test_chain_spanning_several_wavefronts_matches_oracle_and_reference_decisions(name, lanes):
    """lanes_per_chain > 64: one chain per workgroup on lanes/64 wavefronts (each a replica of the scalar logic, partial sums
    exchanged through LDS).  Same bar as the single-wave case: doubles == oracle in the same lane order, decisions == reference."""
    gold = golden_io.load(name)
    case = gold["case"]
    rec = gold["chains"][0]
    spec = model_spec.spec_from_golden(gold, rec)
    if spec["model"] == "pois_glm" and lanes > 256:
        pytest.skip("the GLM kernel is built for workgroups of at most 256 threads")
    s = A.Sampler(spec, chains=3, seed=case["seed"], chain_offset=rec["chain"], lanes_per_chain=lanes)
    li = s.launch_info()
    assert li["lanes_per_chain"] == lanes and li["block_threads"] == lanes and li["grid_blocks"] == 3
    o = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=lanes)
    gs, os_ = run_schedule(s, case["schedule"]), run_schedule(o, case["schedule"])
    assert_chain_equals_oracle(s, 0, o, gs, os_)
    assert s.info()["accepts"][:, 0].tolist() == rec["accepts"]      # the reference's decisions
    assert int(s.diag()["uniforms"][0]) == rec["uniforms"]
    o2 = oracle_lib.OracleChain(spec, case["seed"], rec["chain"] + 2, lanes=lanes)
    assert_chain_equals_oracle(s, 2, o2, gs, run_schedule(o2, case["schedule"]))
    s.close()


@pytest.mark.parametrize("model,n_obs,G", [("normal", 777, 0), ("beta_bern", 1500, 0), ("hier_normal", 900, 6), ("pois_glm", 300, 0)])
def test_many_chains_auto_geometry_vs_oracle(model, n_obs, G):
    """Seeded inputs, auto geometry, ragged N: a sample of chains (first, last, middle) bit-equal to the oracle."""
    data = model_spec.make_data(model, n_obs, 99, G=G or 32, exp=oracle_lib.lib().orc_exp)
    spec = model_spec.build_spec(model, data)
    chains, seed, off = 1000, 4242, 10_000_000_000      # > 2^32 global ids exercise the 64-bit counter words
    s = A.Sampler(spec, chains=chains, seed=seed, chain_offset=off)
    lanes = s.launch_info()["summation_order"]      # (lanes_per_chain, or 1 where the kernel decides against the expression in the reference's order)
    sched = [{"op": "burn", "n": 120}, {"op": "sample", "n": 60, "thin": 3}]
    gs = run_schedule(s, sched)
    for local in (0, 499, 999):
        o = oracle_lib.OracleChain(spec, seed, off + local, lanes=lanes)
        assert_chain_equals_oracle(s, local, o, gs, run_schedule(o, sched))
    s.close()


@pytest.mark.parametrize("n_obs,lanes", [(1, 1), (3, 4), (63, 64), (64, 64), (65, 64), (255, 64), (256, 64), (257, 64), (300, 16), (449, 64), (1023, 256), (130, 128)])
def test_poisson_pass_edges_vs_oracle(n_obs, lanes):
    """The Poisson pass (pairs of rounds through the fused exp/log, the change point resolved per round, a predicated tail): fewer
    observations than lanes, exactly one / two / four rounds, one more and one less, a chain on two and four wavefronts -- chains whose
    change point wanders over the whole range (the wave-uniform 'nobody / everybody adds' rounds and the comparing ones all occur), bit-equal
    to the oracle in the same lane order."""
    data = model_spec.make_data("pois_glm", n_obs, 7 + n_obs, exp=oracle_lib.lib().orc_exp)
    spec = model_spec.build_spec("pois_glm", data)
    s = A.Sampler(spec, chains=6, seed=31, chain_offset=5, lanes_per_chain=lanes)
    assert s.launch_info()["lanes_per_chain"] == lanes
    sched = [{"op": "burn", "n": 60}, {"op": "sample", "n": 40, "thin": 2}]
    gs = run_schedule(s, sched)
    for local in (0, 5):
        o = oracle_lib.OracleChain(spec, 31, 5 + local, lanes=s.launch_info()["summation_order"])      # (16 lanes: the reference's order)
        assert_chain_equals_oracle(s, local, o, gs, run_schedule(o, sched))
    cp = gs[0][:, 8, :]
    assert cp.min() >= 0 and cp.max() <= n_obs - 1
    s.close()


def test_sharding_chunking_and_division_mode_do_not_change_results():
    data = model_spec.make_data("normal", 500, 5)
    spec = model_spec.build_spec("normal", data)
    kw = dict(seed=11, lanes_per_chain=4, block_threads=256)
    whole = A.Sampler(spec, chains=128, **kw)
    whole.burn(77)
    ref = whole.sample(40, 2)
    halves = [A.Sampler(spec, chains=64, chain_offset=o, **kw) for o in (0, 64)]
    parts = []
    for h in halves:
        h.burn(77)
        parts.append(h.sample(40, 2))
    assert np.concatenate(parts, axis=2).tobytes() == ref.tobytes()          # 1 shard == 2 shards
    chunked = A.Sampler(spec, chains=128, steps_per_launch=7, **kw)
    chunked.burn(77)
    assert chunked.sample(40, 2).tobytes() == ref.tobytes()                  # launch chunking
    assert chunked.launch_info()["n_launches"] == 6
    ieee = A.Sampler(spec, chains=128, exact_division=1, **kw)
    ieee.burn(77)
    assert ieee.sample(40, 2).tobytes() == ref.tobytes()                     # hoisted reciprocal == IEEE '/'
    again = A.Sampler(spec, chains=128, **kw)
    again.burn(77)
    assert again.sample(40, 2).tobytes() == ref.tobytes()                    # run-to-run reproducible
    m, sd = whole.moments()
    flat = ref.transpose(1, 0, 2).reshape(2, -1)
    np.testing.assert_allclose(m, flat.mean(axis=1), rtol=1e-12)
    np.testing.assert_allclose(sd, flat.std(axis=1, ddof=1), rtol=1e-10)


def test_draws_fetched_by_slices_launch_by_launch_equal_the_whole_array():
    """amwg_fetch_draws_slices (what sampler.sample() of the JavaScript host is built on): one [kept][len][chains] array per requested
    range of recorded values, copied launch by launch while later launches run -- equal to the [kept][P][chains] array of amwg_sample
    on a twin sampler; ranges may overlap, be empty, or cover everything; a range outside the recorded values is refused."""
    data = model_spec.make_data("hier_normal", 600, 9, G=32)
    spec = model_spec.build_spec("hier_normal", data)
    kw = dict(seed=21, chains=96, lanes_per_chain=8, steps_per_launch=7)
    a, b = A.Sampler(spec, **kw), A.Sampler(spec, **kw)
    a.burn(30); b.burn(30)
    whole = a.sample(45, 2)                                   # 23 kept draws over 7 launches
    b.sample_async(45, 2)
    P = spec["P"]
    parts = b.fetch_draws_slices([(0, 1), (1, 1), (2, P - 2), (0, P), (5, 0), (3, 4)])
    for (base, ln), got in zip([(0, 1), (1, 1), (2, P - 2), (0, P), (5, 0), (3, 4)], parts):
        assert got.shape == (23, ln, 96)
        assert got.tobytes() == np.ascontiguousarray(whole[:, base:base + ln, :]).tobytes()
    assert b.fetch_draws().tobytes() == whole.tobytes()      # and again, in one piece
    b.burn(3)                                                 # launches in between do not invalidate the buffer
    assert b.fetch_draws().tobytes() == whole.tobytes()
    with pytest.raises(A.AmwgError):
        b.fetch_draws_slices([(P - 1, 2)])
    # steps_per_launch = 0: a sample call into the library's buffer is cut into launches of ~32 MB of rows by itself (here 512 steps each)
    spec2 = model_spec.build_spec("normal", model_spec.make_data("normal", 50, 3))
    auto, one = A.Sampler(spec2, chains=4096, seed=5, lanes_per_chain=1), A.Sampler(spec2, chains=4096, seed=5, lanes_per_chain=1, steps_per_launch=65535)
    auto.burn(20); one.burn(20)
    got, want = auto.sample(1200), one.sample(1200)
    assert auto.launch_info()["n_launches"] == 3 and one.launch_info()["n_launches"] == 1
    assert got.tobytes() == want.tobytes()


def test_edge_cases_empty_data_single_chain_thin_larger_than_n():
    spec = model_spec.build_spec("normal", {"x": np.zeros(0)})
    s = A.Sampler(spec, chains=1, seed=3)
    o = oracle_lib.OracleChain(spec, 3, 0, lanes=s.launch_info()["lanes_per_chain"])
    sched = [{"op": "burn", "n": 10}, {"op": "sample", "n": 5, "thin": 9}, {"op": "sample", "n": 0}]
    gs, os_ = run_schedule(s, sched), run_schedule(o, sched)
    assert gs[0].shape == (1, 2, 1) and gs[1].shape == (0, 2, 1)
    assert_chain_equals_oracle(s, 0, o, gs, os_)


def test_set_state_and_convergence_diagnostics():
    """Per-chain starts (amwg_set_state) restart the cached log_post; split-R-hat / ESS equal a numpy restatement."""
    data = model_spec.make_data("normal", 400, 21)
    spec = model_spec.build_spec("normal", data)
    chains, seed = 256, 77
    s = A.Sampler(spec, chains=chains, seed=seed, lanes_per_chain=2)
    rng = np.random.default_rng(5)
    start = np.stack([rng.normal(3, 5, chains), rng.uniform(0.5, 9, chains)])
    s.burn(3)                                    # the cached log_post of the old state must not survive set_state
    s.set_state(start)
    assert s.state().tobytes() == start.tobytes()
    s.burn(40)
    for c in (0, 100, 255):
        # the invariant a stale cache would break: log_post cached on the device == the oracle's log_post at the device's state
        cur = dict(spec, init=s.state()[:, c].tolist())
        assert np.float64(s.diag()["log_post"][c]).tobytes() == np.float64(oracle_lib.OracleChain(cur, seed, c, lanes=2).log_post()).tobytes()
    s.burn(600)
    d = s.sample(200, 2)                         # [100][2][256]
    rhat, ess = s.convergence()
    rows = d.shape[0]
    half = rows // 2
    for p in range(2):
        x = d[:, p, :]
        seqs = np.concatenate([x[:half], x[half:2 * half]], axis=1)       # [half][2C]
        W = seqs.var(axis=0, ddof=1).mean()
        Bn = seqs.mean(axis=0).var(ddof=1)
        var_plus = (half - 1) / half * W + Bn
        np.testing.assert_allclose(rhat[p], np.sqrt(var_plus / W), rtol=1e-10)
        cm = 0.5 * (x[:half].mean(axis=0) + x[half:2 * half].mean(axis=0))
        np.testing.assert_allclose(ess[p], chains * var_plus / cm.var(ddof=1), rtol=1e-9)
        assert 0.98 < rhat[p] < 1.05 and ess[p] > chains
    probs = [0.0, 0.025, 0.25, 0.5, 0.975, 1.0]
    q = s.quantiles(probs)
    for p in range(2):
        np.testing.assert_array_equal(q[p], np.quantile(d[:, p, :].ravel(), probs))      # same order statistics, same interpolation
    s.close()


@pytest.mark.parametrize("n_obs,hyper", [(100000, [2, 2]), (1500, [1, 1]), (33, [0.5, 3.5]), (4097, [40, 3]), (0, [2, 2])])
def test_two_valued_sum_fast_forward_equals_the_term_by_term_pass(n_obs, hyper):
    """Beta-Bernoulli, one lane per chain: the exact fast-forward over binades (csrc/amwg_models.h two_valued_sum) gives the
    same bits as the sequential pass (exact_division = 1, itself pinned against the reference), chain by chain, from
    ordinary and extreme starting points (theta next to 0 and 1, exactly 0 and 1, large |log theta| / tiny log(1-theta))."""
    data = model_spec.make_data("beta_bern", n_obs, 31)
    spec = model_spec.build_spec("beta_bern", data, hyper=hyper)
    chains = 512
    a = A.Sampler(spec, chains=chains, seed=5, lanes_per_chain=1)
    b = A.Sampler(spec, chains=chains, seed=5, lanes_per_chain=1, exact_division=1)
    rng = np.random.default_rng(1)
    start = rng.uniform(0, 1, (1, chains))
    start[0, :12] = [0.0, 1.0, 1e-300, 1e-12, 1 - 1e-12, 0.5, 2.0 ** -30, 1 - 2.0 ** -30, 0.3, 5e-324, np.nextafter(1.0, 0), np.nextafter(0.0, 1)]
    for s in (a, b):
        s.set_state(start)
        s.burn(0)
    la, lb = a.diag()["log_post"], b.diag()["log_post"]
    same = (la.view(np.uint64) == lb.view(np.uint64)) | (np.isnan(la) & np.isnan(lb))
    assert same.all(), (np.where(~same)[0][:5], la[~same][:5], lb[~same][:5], start[0, ~same][:5])
    for s in (a, b):
        s.burn(60)
    da, db = a.sample(40, 2), b.sample(40, 2)
    assert da.tobytes() == db.tobytes()
    assert a.info()["accepts"].tolist() == b.info()["accepts"].tolist()
    assert a.diag()["log_post"].tobytes() == b.diag()["log_post"].tobytes()
    a.close(); b.close()


@pytest.mark.parametrize("groups,lanes,n_obs", [(4, 4, 1000), (4, 8, 999), (8, 64, 3000), (32, 64, 10000), (4, 128, 2000), (6, 64, 900)])
def test_hierarchical_pass_with_lane_constant_groups_equals_the_gathered_pass(groups, lanes, n_obs):
    """When the group labels repeat with the lane stride (g_i = i mod groups, lanes a multiple of groups) every lane of a chain meets
    one group only and HierNormalModel reads its mean once per evaluation instead of gathering it per observation (constant-mean pass,
    ModelConsts::group_lane_const).  Same values, same order: the trajectories must equal the oracle's in that lane order, and the
    term-by-term IEEE schedule (exact_division = 1, which never takes the shortcut); (6, 64) is a layout that must NOT take it."""
    data = model_spec.make_data("hier_normal", n_obs, 321, G=groups)
    spec = model_spec.build_spec("hier_normal", data)
    sched = [{"op": "burn", "n": 110}, {"op": "sample", "n": 40, "thin": 4}]
    s = A.Sampler(spec, chains=6, seed=8, chain_offset=3, lanes_per_chain=lanes)
    e = A.Sampler(spec, chains=6, seed=8, chain_offset=3, lanes_per_chain=lanes, exact_division=1)
    gs, ge = run_schedule(s, sched), run_schedule(e, sched)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(gs, ge))
    assert s.state().tobytes() == e.state().tobytes()
    for local in (0, 5):
        o = oracle_lib.OracleChain(spec, 8, 3 + local, lanes=s.launch_info()["summation_order"])      # (the sweep kernel: the reference's order)
        assert_chain_equals_oracle(s, local, o, gs, run_schedule(o, sched))
    s.close()
    e.close()


@pytest.mark.parametrize("model,n_obs,G,chains", [("normal", 2000, 0, 4096), ("hier_normal", 1200, 6, 300), ("beta_bern", 3000, 0, 2000)])
def test_autotuned_geometry_leaves_no_trace_in_the_chains(model, n_obs, G, chains):
    """lanes_per_chain = AMWG_LANES_AUTOTUNE (-2): every lane count that fits is run for a few steps at construction and timed; the chain
    state is saved before and restored after, so the tuned sampler must produce exactly what a sampler constructed with the chosen lane
    count produces (same draws, counters, uniforms consumed)."""
    data = model_spec.make_data(model, n_obs, 17, G=G or 32, exp=oracle_lib.lib().orc_exp)
    spec = model_spec.build_spec(model, data)
    tuned = A.Sampler(spec, chains=chains, seed=5, chain_offset=7, lanes_per_chain=-2)
    cands = tuned.tuning()
    lanes = tuned.launch_info()["lanes_per_chain"]
    assert len(cands) >= 3 and lanes in [c[0] for c in cands] and all(ms > 0 for _, ms in cands)
    best = min(ms for _, ms in cands)
    one = [ms for l, ms in cands if l == 1]
    assert dict(cands)[lanes] <= 1.12 * best or (lanes == 1 and one and one[0] <= 1.12 * best)
    plain = A.Sampler(spec, chains=chains, seed=5, chain_offset=7, lanes_per_chain=lanes)
    sched = [{"op": "burn", "n": 60}, {"op": "sample", "n": 30, "thin": 3}]
    gt, gp = run_schedule(tuned, sched), run_schedule(plain, sched)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(gt, gp))
    it, ip = tuned.info(), plain.info()
    for k in it:
        assert it[k].tobytes() == ip[k].tobytes(), k
    assert tuned.diag()["uniforms"].tobytes() == plain.diag()["uniforms"].tobytes()
    tuned.close()
    plain.close()


def test_run_totals_survive_launches_longer_than_their_16_bit_launch_counters():
    """A launch counts accepted / evaluated proposals in 16-bit fields of one LDS word per component and adds them to the 32-bit totals when
    it ends; the host therefore cuts a burn() into launches of at most 65 535 steps.  70 000 steps of an unbounded parameter: evaluated ==
    steps, accepted <= evaluated, two launches."""
    spec = model_spec.build_spec("normal", model_spec.make_data("normal", 16, 3))
    s = A.Sampler(spec, chains=8, seed=2, lanes_per_chain=1)
    s.burn(70_000)
    info = s.info()
    assert s.launch_info()["n_launches"] == 2
    assert np.all(info["inbounds"][0] == 70_000)                      # mu is unbounded: every proposal is evaluated
    assert np.all(info["accepts"] <= info["inbounds"]) and np.all(info["accepts"][0] > 20_000)
    o = oracle_lib.OracleChain(spec, 2, 0, lanes=1)
    o.burn(70_000)
    assert info["accepts"][:, 0].tolist() == o.info()["accepts"].tolist() and s.state()[:, 0].tolist() == o.state().tolist()
    s.close()


def test_binary_parameter_on_a_built_in_family_is_refused():
    """The built-in kernels are compiled without the BinaryStepper branch; through the C ABI a binary-typed parameter must be an error, not
    a silent Metropolis update (mcmc.js:753-767 is what the reference runs for it)."""
    data = model_spec.make_data("normal", 50, 3)
    spec = model_spec.build_spec("normal", data)
    spec["params"][0] = dict(spec["params"][0], type="binary")
    with pytest.raises(A.AmwgError, match="binary"):
        A.Sampler(spec, chains=4, seed=1)


def _group_local_case(name):
    gold = golden_io.load(name)
    return gold, gold["case"], gold["chains"][0]


@pytest.mark.parametrize("name", ["hier_small", "cfg4_full", "hier_hyper"])      # hier_hyper: 5 groups (not a power of two), 300 observations
def test_group_local_sweep_equals_its_oracle_and_takes_the_reference_decisions(name):
    """amwg_options::group_local (hierarchical family): the lane-parallel sweep over theta -- all G proposals of a step evaluated in one pass,
    stream positions resolved on the scalar unit -- against its sequential restatement in oracle/amwg_oracle.c (gl_*): every double bit for
    bit; and against the seeded run of the unmodified reference: every accept decision, adaptation step and uniform count."""
    gold, case, rec = _group_local_case(name)
    spec = model_spec.spec_from_golden(gold, rec)
    s = A.Sampler(spec, chains=5, seed=case["seed"], chain_offset=rec["chain"], group_local=1)
    assert s.launch_info()["lanes_per_chain"] == 64
    o = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=64, group_local=True)
    gs, os_ = run_schedule(s, case["schedule"]), run_schedule(o, case["schedule"])
    assert_chain_equals_oracle(s, 0, o, gs, os_)
    assert s.info()["accepts"][:, 0].tolist() == rec["accepts"]      # the reference's decisions
    assert s.info()["batch_count"][:, 0].tolist() == rec["batch_count"]
    assert int(s.diag()["uniforms"][0]) == rec["uniforms"]
    o3 = oracle_lib.OracleChain(spec, case["seed"], rec["chain"] + 3, lanes=64, group_local=True)
    assert_chain_equals_oracle(s, 3, o3, gs, run_schedule(o3, case["schedule"]))
    s.close()


def test_group_local_sweep_with_bounded_integer_and_non_adapting_components():
    """The sweep's resolution of stream positions assumes every proposal falls inside its bounds (and so draws an accept uniform) and
    repairs the assumption where it was wrong: tight bounds on theta make that the common case here; an int-typed theta, per-component
    options, launches of 7 steps and a stop / start of the adaptation ride along."""
    data = model_spec.make_data("hier_normal", 777, 31, G=16)
    spec = model_spec.build_spec("hier_normal", data)
    spec["params"][0] = dict(spec["params"][0], lower=3.0, upper=7.5, init=[5.25] * 16)
    spec["init"] = [5.25] * 16 + list(spec["init"][16:])
    for i, o in enumerate(spec["comp_opts"]):
        o.update(batch_size=7 + (i % 3), prop_log_scale=0.8, is_adapting=(i % 5 != 2))
    sched = [{"op": "burn", "n": 61}, {"op": "stop"}, {"op": "sample", "n": 20, "thin": 2}, {"op": "start"}, {"op": "sample", "n": 45, "thin": 3}]
    for typ in ("real", "int"):
        spec["params"][0] = dict(spec["params"][0], type=typ)
        s = A.Sampler(spec, chains=3, seed=99, chain_offset=7, group_local=1, steps_per_launch=7)
        o = oracle_lib.OracleChain(spec, 99, 8, lanes=64, group_local=True)
        gs, os_ = run_schedule(s, sched), run_schedule(o, sched)
        assert_chain_equals_oracle(s, 1, o, gs, os_)
        inb = s.info()["inbounds"][:16, 1]
        assert (inb < 126).any()          # some proposals did fall outside
        s.close()


def _shuffled_hier(n_obs, G, seed, sizes=None):
    """A hierarchical data set whose labels are NOT i mod G: group sizes as given (or random), observations in random order."""
    rng = np.random.default_rng(seed)
    if sizes is None:
        sizes = rng.multinomial(n_obs, rng.dirichlet(np.ones(G) * 2.0))
    g = np.repeat(np.arange(G), sizes).astype(np.int32)
    rng.shuffle(g)
    theta = rng.normal(5.0, 3.0, G)
    y = theta[g] + rng.normal(0.0, 2.0, g.size)
    return {"x": y, "g": g, "G": G}


@pytest.mark.parametrize("n_obs,G,sizes", [(1000, 5, None), (777, 13, None), (640, 64, None), (300, 2, None), (900, 3, [800, 99, 1]), (40, 4, [0, 25, 15, 0]),
                                           (2000, 32, None), (130, 64, [3] * 2 + [2] * 62)])
def test_group_local_any_labels_any_group_count_equals_its_oracle(n_obs, G, sizes):
    """Round 4: the group-local kernel takes any labels and any G <= 64 -- the host deals the wavefront's lanes to the groups in aligned
    power-of-two blocks and lays the data out lane-major (amwg_core.hip gl_layout; restated in the oracle).  Ragged designs: groups of
    very different sizes, empty groups, two groups, 64 groups, fewer observations than lanes; chains 0 and 2 of 3, every double."""
    data = _shuffled_hier(n_obs, G, 100 + n_obs + G, sizes)
    spec = model_spec.build_spec("hier_normal", data)
    s = A.Sampler(spec, chains=3, seed=77, chain_offset=11, group_local=1)
    assert s.launch_info()["lanes_per_chain"] == 64
    sched = [{"op": "burn", "n": 130}, {"op": "sample", "n": 60, "thin": 3}]
    gs = run_schedule(s, sched)
    for local in (0, 2):
        o = oracle_lib.OracleChain(spec, 77, 11 + local, lanes=64, group_local=True)
        assert_chain_equals_oracle(s, local, o, gs, run_schedule(o, sched))
    s.close()


def test_group_local_preconditions_are_enforced():
    data = model_spec.make_data("hier_normal", 300, 5, G=6)
    data = dict(data, G=65, g=np.arange(300, dtype=np.int32) % 65)      # 65 groups: more than the lanes of a wavefront
    spec = model_spec.build_spec("hier_normal", data, G=65)
    with pytest.raises(A.AmwgError, match="1 to 64 groups"):
        A.Sampler(spec, chains=2, seed=1, group_local=1)
    with pytest.raises(A.AmwgError, match="hierarchical"):
        A.Sampler(model_spec.build_spec("normal", model_spec.make_data("normal", 100, 5)), chains=2, seed=1, group_local=1)


@pytest.mark.parametrize("n_obs,G,chains,theta", [(10_000, 32, 300, None), (1_000, 8, 130, None), (640, 64, 70, None), (257, 2, 65, None),
                                                  (2_000, 32, 130, "bounded"), (1_000, 16, 70, "int"), (1_500, 8, 70, "shifted"), (900, 16, 66, "tiny")])
def test_lane_local_reevaluation_equals_the_full_evaluation(n_obs, G, chains, theta):
    """Hierarchical family on a wavefront per chain (labels that repeat with the lane stride), over a schedule with short launches, a stop / start of the
    adaptation, thinning, and states overwritten from the host in between (every cached sum is then stale).
    * options.full_evaluation = 1 (every evaluation passes over all the data) and = 2 (the row layout: only the lanes whose sum an update can have changed are
      re-formed, amwg_models.h lane_sum_rows) evaluate the expression in the 64-lane order: they must agree with each other in EVERY bit of every chain.
    * the default decides from certified sums against the expression in the REFERENCE's order (one running sum; amwg_models.h reference_order): it must agree in
      every bit -- draws, counters, proposal scales, uniforms, the cached log_post -- with the same sampler at ONE lane per chain (the reference's order by
      construction, pinned to the reference goldens elsewhere), with its bounds widened 2^12-fold (the expression often) and 2^40-fold (always: every decision of
      the 64-lane kernel is then made by reference_order itself).
    * the two groups agree in everything but the last bits of log_post (two summation orders of the same terms)."""
    data = model_spec.make_data("hier_normal", n_obs, 77, G=G)
    spec = model_spec.build_spec("hier_normal", data)
    # theta with bounds some proposals fall outside of, or of integer type: the sweep kernel then walks the parameter update by update (an update
    # that draws no accept uniform cannot be drawn ahead by the lane-parallel resolution), still re-forming only the lanes an update changes
    if theta == "bounded":
        spec["params"][0] = dict(spec["params"][0], lower=-0.5, upper=6.5)
    elif theta == "int":
        spec["params"][0] = dict(spec["params"][0], type="int", lower=-3.0, upper=8.0, init=[float(round(v)) for v in spec["params"][0]["init"]])
        spec["init"] = [v for p in spec["params"] for v in p["init"]]
    # labels shifted against the lanes (lane j holds a term of theta_j's prior but the observations of another group: its sum depends on TWO components),
    # or an observation outside the range the 4-operation quotient needs (IEEE division everywhere, no fast pass): the sweep is drawn ahead, its sums cannot be prepared, and the updates take their proposals from the lanes
    kw = {}
    if theta == "shifted":
        data = dict(data, g=((np.arange(n_obs) + 3) % G).astype(np.int32))
        spec = model_spec.build_spec("hier_normal", data)
    elif theta == "tiny":
        x = np.array(data["x"], dtype=np.float64)
        x[5] = 1e-250
        data = dict(data, x=x)
        spec = model_spec.build_spec("hier_normal", data)
    mk = lambda full, shift=0: A.Sampler(spec, chains=chains, seed=4, chain_offset=9, lanes_per_chain=64, steps_per_launch=7, full_evaluation=full, test_bound_shift=shift, **kw)
    # (2: the row layout and the sweep prefetch like 0, but every sweep's accept tests decided update by update instead of all at once from the entries' local
    # differences with their rounding bound -- the path a sweep takes when a uniform falls inside that bound, some 1e-8 of the sweeps otherwise)
    # (test_bound_shift = 12: the rounding bounds of the all-at-once sweep decisions and of mu's early rejection made 4096 times wider -- a good share of the
    # sweeps then meets a uniform inside the bound and is walked update by update, mixed with sweeps that are not)
    # (22: bounds of ~0.1 -- the WIDE regime of the certified test, csrc/amwg_kernel.h certified_test_wide: hopeless proposals rejected without an exponential, the
    # rest from exp_v8(dA -+ eta), a good share by the expression)
    a, b, c2, c3, c4, c5 = mk(0), mk(1), mk(2), mk(0, 12), mk(0, 40), mk(0, 22)
    one = A.Sampler(spec, chains=chains, seed=4, chain_offset=9, lanes_per_chain=1, steps_per_launch=7, full_evaluation=1, **kw)
    assert a.launch_info()["lds_bytes"] != b.launch_info()["lds_bytes"]      # the row layout (tile + term rows) is in use on one side only
    assert a.launch_info()["kernel"] == c3.launch_info()["kernel"] == c4.launch_info()["kernel"] and a.launch_info()["kernel"].startswith("amwg_sweep_kernel_cert<HierNormalModel")
    assert c2.launch_info()["kernel"].startswith("amwg_sweep_kernel<HierNormalModel") and b.launch_info()["kernel"].startswith("amwg_step_kernel<HierNormalModel,64")
    assert [s.launch_info()["summation_order"] for s in (a, c3, c4, c5, one, b, c2)] == [1, 1, 1, 1, 1, 64, 64]
    assert one.launch_info()["lanes_per_chain"] == 1
    outs = []
    for s in (a, c3, c4, c5, one, b, c2):
        seq = [s.sample(40, 1)]
        s.burn(33)
        s.set_adapting(False)
        seq.append(s.sample(25, 4))
        s.set_adapting(True)
        st = s.state()
        st[:, ::3] = np.random.default_rng(11).normal(5.0, 2.0, st[:, ::3].shape)
        st[-1] = np.abs(st[-1]) + 0.5                                          # sigma stays inside its bounds
        if theta == "bounded":
            st[:G] = np.clip(st[:G], -0.5, 6.5)
        elif theta == "int":
            st[:G] = np.clip(np.round(st[:G]), -3.0, 8.0)
        s.set_state(st)
        s.burn(50)
        seq.append(s.sample(30, 2))
        outs.append((seq, s.info(), s.diag(), s.state()))
    _same_chains(outs[0], outs[1:5], log_post_bits=True)       # default == bounds widened (three ways) == ONE lane per chain
    _same_chains(outs[5], outs[6:], log_post_bits=True)        # the two 64-lane-order evaluations
    _same_chains(outs[0], outs[5:6], log_post_bits=False)      # ... and across: everything but the last bits of log_post
    for s in (a, b, c2, c3, c4, c5, one):
        s.close()


def _same_chains(ref, others, log_post_bits):
    (sa, ia, da, sta) = ref
    for (sb, ib, db, stb) in others:
        for x, y in zip(sa, sb):
            assert x.tobytes() == y.tobytes()
        for k in ia:
            assert ia[k].tobytes() == ib[k].tobytes(), k
        assert da["uniforms"].tobytes() == db["uniforms"].tobytes()
        if log_post_bits:
            assert da["log_post"].tobytes() == db["log_post"].tobytes()
        else:
            fin = np.isfinite(da["log_post"])
            assert np.array_equal(fin, np.isfinite(db["log_post"])) and np.allclose(da["log_post"][fin], db["log_post"][fin], rtol=1e-11, atol=0)
        assert sta.tobytes() == stb.tobytes()


@pytest.mark.parametrize("n_obs,chains,steps", [(500, 1024, 300), (3000, 512, 120), (449, 260, 200), (61, 256, 200)])
def test_certified_decisions_of_the_poisson_family_equal_the_expression_in_every_update(n_obs, chains, steps):
    """Poisson GLM + integer change point, 16 lanes per chain (four chains to a wavefront): by default the accept test is decided from  prior + sum eta y - sum
    e^eta - sum lfactorial(y)  -- no logarithm of the exponential, the four chains of a wavefront sharing every row they read -- and a bound on its distance
    from the reference's expression IN THE REFERENCE'S ORDER (csrc/amwg_models.h PoisGlmModel::log_post_approx, reference_order).  Every bit of every chain --
    draws, counters, proposal scales, uniforms, the cached log_post -- must agree with the same sampler at ONE lane per chain (the reference's order by
    construction), with the bound widened 2^14- and 2^40-fold as well (updates fall back to the expression often / always: at 2^40 every decision is made by
    reference_order itself); options.full_evaluation = 1 (the expression in the 16-lane order in every update) agrees in everything but the last bits of
    log_post.  Chain counts that leave a wavefront partly filled, fewer observations than lanes."""
    data = model_spec.make_data("pois_glm", n_obs, 123, exp=oracle_lib.lib().orc_exp)
    spec = model_spec.build_spec("pois_glm", data)
    mk = lambda full, shift=0: A.Sampler(spec, chains=chains, seed=21, chain_offset=2, lanes_per_chain=16, steps_per_launch=9, full_evaluation=full, test_bound_shift=shift)
    outs = []
    one = A.Sampler(spec, chains=chains, seed=21, chain_offset=2, lanes_per_chain=1, steps_per_launch=9, full_evaluation=1)
    for s in (mk(0), mk(0, 14), mk(0, 40), mk(0, 18), one, mk(1)):      # (18: bounds of ~0.3, the wide regime of the certified test)
        assert s.launch_info()["lanes_per_chain"] == (1 if s is one else 16)
        seq = [s.sample(steps // 3, 2)]
        s.burn(steps // 3)
        s.set_adapting(False)
        seq.append(s.sample(steps // 6, 1))
        s.set_adapting(True)
        s.burn(steps // 6)
        outs.append((seq, s.info(), s.diag(), s.state()))
        s.close()
    _same_chains(outs[0], outs[1:5], log_post_bits=True)
    _same_chains(outs[0], outs[5:], log_post_bits=False)


@pytest.mark.parametrize("n_obs,chains,steps,hyper", [(1000, 4096, 400, None), (777, 1024, 300, None), (17, 512, 300, None), (1000, 16384, 600, [0.0, 100.0, 0.0, 1.0])])
def test_certified_decisions_equal_the_expression_in_every_update(n_obs, chains, steps, hyper):
    """Normal family, one lane per chain (the reference's order): by default the accept test is decided from  prior + n c - sum (x - mu)^2 / den  and a bound on
    its distance from the reference's term-by-term expression (csrc/amwg_kernel.h "certified decisions"); options.full_evaluation = 1 evaluates the expression
    in every update.  The two must agree in EVERY bit of every chain -- draws, counters, proposal scales, uniforms, and the cached log_post, which is the
    expression's on both sides -- over short launches (a launch ends by evaluating the expression), a stop / start of the adaptation and a state overwritten
    from the host.  test_bound_shift widens the bound 2^14- and 2^40-fold: updates then fall back to the expression often / always.  The last case (data
    45 sd wide against a sigma pinned below 1: |log_post| ~ 1e6) has a bound so wide by itself that tens of its 2e7 decisions fall inside it."""
    data = model_spec.make_data("normal", n_obs, 31)
    if hyper is not None:
        data = dict(data, x=3.0 + 22.5 * (np.array(data["x"], dtype=np.float64) - 3.0))
    spec = model_spec.build_spec("normal", data, hyper=hyper)
    if hyper is not None:
        spec["params"][1] = dict(spec["params"][1], upper=1.0)
    mk = lambda full, shift=0, suff=0: A.Sampler(spec, chains=chains, seed=8, chain_offset=3, lanes_per_chain=1, steps_per_launch=13, full_evaluation=full, test_bound_shift=shift, sufficient_statistics=suff)
    variants = [mk(0), mk(1), mk(0, 14), mk(0, 40), mk(0, 22)] if hyper is None else [mk(0), mk(1), mk(0, 9)]      # (22 / 9: bounds of ~0.1, the wide regime of the certified test)
    # (round 6, last day: the pass takes the wavefront's 64 means through a scratch line and the scalar cache -- csrc/amwg_pass.h --; AMWG_WAVE_SCRATCH=0 at construction
    # keeps the v_readlane broadcast: the same sums in the same order)
    os.environ["AMWG_WAVE_SCRATCH"] = "0"
    try:
        variants.append(mk(0))
    finally:
        del os.environ["AMWG_WAVE_SCRATCH"]
    # (round 6, opt-in third tier -- options.sufficient_statistics: the cheap value from SS + n (xbar - mu)^2, no pass over the data; same bound, same fallback, same bits)
    variants += [mk(0, 0, 1), mk(0, 14, 1)]
    outs = []
    for s in variants:
        assert s.launch_info()["lanes_per_chain"] == 1
        seq = [s.sample(steps // 4, 3)]
        s.burn(steps // 2)
        s.set_adapting(False)
        seq.append(s.sample(steps // 8, 1))
        s.set_adapting(True)
        st = s.state()
        st[0, ::3] += 0.25
        s.set_state(st)
        s.burn(steps // 8)
        seq.append(s.sample(20, 2))
        outs.append((seq, s.info(), s.diag(), s.state()))
        s.close()
    (sa, ia, da, sta) = outs[0]
    for (sb, ib, db, stb) in outs[1:]:
        for x, y in zip(sa, sb):
            assert x.tobytes() == y.tobytes()
        for k in ia:
            assert ia[k].tobytes() == ib[k].tobytes(), k
        assert da["uniforms"].tobytes() == db["uniforms"].tobytes() and da["log_post"].tobytes() == db["log_post"].tobytes()
        assert sta.tobytes() == stb.tobytes()
