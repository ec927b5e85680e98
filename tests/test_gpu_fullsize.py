"""-m gpu: BASELINE.json configs 2 and 3 at FULL size (65 536 / 262 144 chains, 1e4 / 1e5 observations) through size-independent
properties -- the oracle finishes single chains of these sizes only (tests/golden/cfg*_full.json pin two chains each bit for bit):
  * the pooled draws sample the analytic posterior (means within 0.1 %, standard deviations within 3 %; north_star asks for 1 %),
  * Roberts-Rosenthal adaptation drives every component's acceptance rate to the 0.44 target (mcmc.js:543),
  * every chain is a valid run: finite state, accepted <= evaluated <= steps, uniforms consumed >= 3 per in-bounds update,
  * chains are independent draws of the same process: split-R-hat ~ 1."""
import math

import numpy as np
import pytest

import amwg_ctypes as A
import model_spec

pytestmark = pytest.mark.gpu


def _run(spec, chains, burn, n, thin):
    s = A.Sampler(spec, chains=chains, seed=20260925)      # auto geometry: reference order where it is priced within 12 %
    s.burn(burn)
    acc0, inb0 = s.info()["accepts"].copy(), s.info()["inbounds"].copy()
    s.sample_async(n, thin)
    s.sync()
    return s, acc0, inb0


def test_cfg2_full_size_samples_the_analytic_posterior_and_adapts_to_the_target():
    data = model_spec.make_data("normal", 10_000, 20260925)
    spec = model_spec.build_spec("normal", data)
    x = np.asarray(data["x"])
    n, xbar, s2 = x.size, x.mean(), x.var(ddof=1)
    C, steps = 65_536, 600
    smp, acc0, inb0 = _run(spec, C, 1500, steps, 6)
    assert smp.launch_info()["lanes_per_chain"] == 1                      # the default geometry of cfg2 is the reference's summation order
    mean, sd = smp.moments()
    # mu | data ~ t_{n-1}(xbar, s^2/n) (the N(0,100) prior is flat at this scale); sigma^2 | data ~ Inv-Gamma((n-1)/2, (n-1) s^2 / 2) under a uniform prior on sigma
    assert abs(mean[0] - xbar) < 1e-3 * abs(xbar)
    assert abs(sd[0] - math.sqrt(s2 / n * (n - 1) / (n - 3))) < 0.03 * math.sqrt(s2 / n)
    e_sigma = math.sqrt(s2 * (n - 1) / 2) * math.exp(math.lgamma((n - 2) / 2) - math.lgamma((n - 1) / 2))
    assert abs(mean[1] - e_sigma) < 1e-3 * e_sigma
    assert abs(sd[1] - math.sqrt(s2 / (2 * n))) < 0.03 * math.sqrt(s2 / (2 * n))
    info = smp.info()
    rate = (info["accepts"] - acc0) / float(steps)
    assert np.all(np.abs(rate.mean(axis=1) - 0.44) < 0.02), rate.mean(axis=1)      # batch adaptation towards target_accept_rate
    assert np.all(info["accepts"] <= info["inbounds"]) and np.all(info["inbounds"] <= 1500 + steps)
    assert np.all(np.isfinite(smp.state()))
    un = smp.diag()["uniforms"].astype(np.int64)
    assert np.all(un >= 3 * info["inbounds"].sum(axis=0))                  # >= 2 per rnorm pass + 1 per accept test; +1 per step for the shuffle
    rhat, ess = smp.convergence()
    assert np.all(np.abs(rhat - 1) < 0.01) and np.all(ess > 0.5 * C)
    smp.close()


def test_cfg3_full_size_samples_the_conjugate_posterior():
    data = model_spec.make_data("beta_bern", 100_000, 20260925)
    spec = model_spec.build_spec("beta_bern", data)
    x = np.asarray(data["x"])
    a, b = 2 + x.sum(), 2 + x.size - x.sum()                               # Beta(2,2) prior (README.md:149-164) => Beta(2 + sum x, 2 + n - sum x)
    C, steps = 262_144, 400
    smp, acc0, _ = _run(spec, C, 1200, steps, 4)
    assert smp.launch_info()["lanes_per_chain"] == 1
    mean, sd = smp.moments()
    assert abs(mean[0] - a / (a + b)) < 1e-3 * a / (a + b)
    assert abs(sd[0] - math.sqrt(a * b / ((a + b) ** 2 * (a + b + 1)))) < 0.03 * math.sqrt(a * b / ((a + b) ** 2 * (a + b + 1)))
    rate = (smp.info()["accepts"] - acc0) / float(steps)
    assert abs(rate.mean() - 0.44) < 0.02
    q = smp.quantiles([0.025, 0.5, 0.975])[0]
    from scipy.stats import beta as beta_dist
    want = beta_dist.ppf([0.025, 0.5, 0.975], a, b)
    assert np.all(np.abs(q - want) < 2e-4)                                 # posterior sd is 1.4e-3: quantiles to a seventh of it
    smp.close()


def test_cfg4_per_gpu_size_hierarchical_posterior_properties():
    """cfg4 as one GPU of eight sees it: 2 048 chains, 1e4 observations in 32 groups, 34 components (64 lanes per chain, the group labels
    repeat with the lane stride: constant-mean pass).  No closed form with sigma unknown; the properties: every component adapts to the
    0.44 target, chains agree (split-R-hat), the group means sit at the group averages (prior sd 10 vs sigma/sqrt(n_g) = 0.11: shrinkage
    < 0.02 %), sigma at the pooled within-group sd."""
    data = model_spec.make_data("hier_normal", 10_000, 20260925, G=32)
    spec = model_spec.build_spec("hier_normal", data)
    y, g = np.asarray(data["x"]), np.asarray(data["g"])
    C, steps = 2_048, 500
    smp, acc0, _ = _run(spec, C, 2000, steps, 5)
    assert smp.launch_info()["lanes_per_chain"] == 64
    mean, sd = smp.moments()
    gm = np.array([y[g == k].mean() for k in range(32)])
    ng = np.array([(g == k).sum() for k in range(32)])
    within = math.sqrt(sum(((y[g == k] - gm[k]) ** 2).sum() for k in range(32)) / (y.size - 32))
    assert np.all(np.abs(mean[:32] - gm) < 0.25 * within / np.sqrt(ng))            # a quarter of a posterior sd
    assert np.all(np.abs(sd[:32] - within / np.sqrt(ng)) < 0.05 * within / np.sqrt(ng))
    assert abs(mean[33] - within) < 0.002 * within and abs(mean[32] - gm.mean()) < 4 * 10 / math.sqrt(32) * 0.25
    rate = (smp.info()["accepts"] - acc0) / float(steps)
    assert np.all(np.abs(rate.mean(axis=1) - 0.44) < 0.03), rate.mean(axis=1)
    rhat, ess = smp.convergence()
    assert np.all(np.abs(rhat - 1) < 0.02), rhat
    smp.close()


def test_cfg5_full_size_chains_agree_and_sit_on_the_generating_values():
    """cfg5 at full size (8 real coefficients + an int change point, 5e4 observations, 64 lanes per chain) on 1 024 chains.  The reference needs
    ~4 h for ONE chain of this length, so there is no reference sample to test against at this size -- that comparison is made at N = 500
    (tests/test_gpu_moments.py: eight reference chains of 2e4 steps each, decision for decision and as a two-sample test) and the full-size
    trajectory is pinned by the cfg5_full golden (both chain ids, 300 steps).  What full size adds: the posterior the chains settle on.  With
    5e4 observations it is tight around the generating values (oracle/synth.js glm: beta_true, cp_true = 20 000), so the pooled mean of every
    coefficient must sit within a few posterior sds of the truth, all chains must agree (split-R-hat) and every component must adapt to the
    0.44 target."""
    data = model_spec.make_data("pois_glm", 50_000, 20260925, exp=A.lib().amwg_exp)
    spec = model_spec.build_spec("pois_glm", data)
    smp = A.Sampler(spec, chains=1_024, seed=20260925, lanes_per_chain=64, steps_per_launch=100)      # (the lane count cfg5 is timed at; launches of ~0.5 s)
    smp.burn(2500)
    acc0 = smp.info()["accepts"].copy()
    smp.sample_async(500, 5)
    smp.sync()
    st = smp.state()
    assert np.all(np.isfinite(st)) and np.all(st[8] >= 0) and np.all(st[8] <= 49_999) and np.all(st[8] == np.round(st[8]))
    mean, sd = smp.moments()
    beta_true = np.array([0.5, 0.2, -0.1, 0.05, 0.1, -0.2, 0.15, 0.3])
    assert np.all(sd[:8] < 0.02), sd                                                # 5e4 observations: posterior sds of a few 1e-3
    assert np.all(np.abs(mean[:8] - beta_true) < 5 * sd[:8] + 0.01), ((mean[:8] - beta_true) / sd[:8]).round(2)
    assert abs(mean[8] - 20_000) < 5 * sd[8] + 50, (mean[8], sd[8])
    rhat, _ = smp.convergence()
    assert np.all(np.abs(rhat[:8] - 1) < 0.05), rhat
    rate = (smp.info()["accepts"] - acc0) / 500.0
    assert np.all(np.abs(rate[:8].mean(axis=1) - 0.44) < 0.04), rate.mean(axis=1)
    smp.close()


@pytest.mark.parametrize("workload", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_chains_taken_out_of_the_timed_full_size_sampler_match_the_reference(workload):
    """The geometry bench.py times (65 536 chains x 256 workgroups for cfg2, 262 144 for cfg3, 2 048 per GPU on 64 lanes for cfg4, 8 192 for
    cfg5), not a 3-chain stand-in: the golden schedule is run ON the full-size sampler and the golden's chain ids (first and last of the
    job; for cfg4 the last one lives in the 8th shard) are read out of it -- bit-identical draws / sums / state with one lane per chain,
    identical decisions, adaptation state and uniform counts with more (mcmc.js:1020-1027, 517-553)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    if workload == "cfg2":
        spec, chains = bench.normal_spec(), bench.CHAINS_PER_GPU
    else:
        spec, chains = bench.other_spec(workload, A.lib().amwg_exp), bench.OTHER_WORKLOADS[workload][2]
    mk = lambda off: A.Sampler(spec, chains=chains, seed=bench.SEED, chain_offset=off, steps_per_launch=100)
    probe = mk(0)
    lanes = probe.launch_info()["lanes_per_chain"]
    probe.close()
    report, s = bench.timed_geometry_parity(A, spec, workload, chains, mk, lanes)
    s.close()
    assert report["from_timed_sampler"] and report["chains_in_sampler"] == chains
    for k, v in report.items():
        if k.endswith("_identical"):
            assert v is True, (k, report)
    if workload in ("cfg2", "cfg3"):
        assert lanes == 1 and report["draws_bit_identical"] and report["final_state_bit_identical"]
