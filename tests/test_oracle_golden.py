"""Pins the CPU oracle (oracle/amwg_oracle.c) against the UNMODIFIED reference.

tests/golden/*.json were produced by oracle/gen_golden.js: the reference sampler
(mcmc.js + distributions.js) run under Node with Math.random replaced by the seeded
Philox twin.  With lanes=1 (the reference's sequential summation order) the oracle must
reproduce every recorded number bit for bit.  With lanes>1 (the HIP kernel's summation
order) every accept decision must still be identical (north_star: "same seed => bit-identical
integer accept counts") and draws agree to ~1e-12 relative.
"""
import numpy as np
import pytest

import golden_io
import model_spec
import oracle_lib

CASES = ["cfg1_heights", "normal_n1000", "cfg2_full", "normal_opts", "beta_bern_n2000", "cfg3_full", "hier_small",
         "cfg4_full", "glm_small", "cfg5_full", "normal_hyper", "beta_bern_hyper", "beta_bern_hyper2", "hier_hyper", "glm_hyper",
         "cfg4_theta_bounded", "cfg4_theta_int"]      # (configs[3] with a bounded / an integer theta: updates that draw no accept uniform, rounded proposals)


def run_schedule(chain, schedule):
    segs = []
    thin = 1
    for seg in schedule:
        if seg["op"] == "burn":
            chain.burn(seg["n"])
        elif seg["op"] == "stop":
            chain.set_adapting(False)
        elif seg["op"] == "start":
            chain.set_adapting(True)
        elif seg["op"] == "sample":
            thin = seg.get("thin", thin)
            segs.append(chain.sample(seg["n"], thin))
    return segs


def check_chain(gold, rec, lanes):
    spec = model_spec.spec_from_golden(gold, rec)
    ch = oracle_lib.OracleChain(spec, gold["case"]["seed"], rec["chain"], lanes=lanes)
    segs = run_schedule(ch, gold["case"]["schedule"])
    info = ch.info()
    exact = lanes == 1
    # integer state: identical in every summation order
    assert info["accepts"].tolist() == rec["accepts"]
    assert info["inbounds"].tolist() == rec["inbounds"]
    assert info["batch_count"].tolist() == rec["batch_count"]
    assert info["acceptance_count"].tolist() == rec["acceptance_count"]
    assert info["iterations_since_adaption"].tolist() == rec["iterations_since_adaption"]
    assert ch.uniforms() == rec["uniforms"]
    assert ch.named_order().tolist() == rec["named_order"]
    # adaptation is driven by the integer counts only => exact in every order
    assert info["prop_log_scale"].tolist() == rec["prop_log_scale"]
    for got, want in zip(segs, rec["samples"]):
        assert got.shape[0] == want["kept"]
        w = np.array(want["draws"], dtype=np.float64).reshape(-1, got.shape[1])
        g = got[: w.shape[0]]
        if exact:
            assert g.tobytes() == w.tobytes()
            # running sum over ALL kept draws, same sequential order as the harness
            s = np.zeros(got.shape[1])
            for t in range(got.shape[0]):
                s = s + got[t]
            assert s.tolist() == want["sum"]
        else:
            np.testing.assert_allclose(g, w, rtol=1e-11, atol=0)
    if exact:
        assert ch.state().tolist() == rec["final_state"]
        assert ch.log_post() == rec["log_post"]
        assert ch.log_post_unhoisted() == rec["log_post"]
    else:
        np.testing.assert_allclose(ch.state(), rec["final_state"], rtol=1e-11)


@pytest.mark.parametrize("name", CASES)
def test_oracle_bit_exact_vs_reference(name):
    gold = golden_io.load(name)
    for rec in gold["chains"]:
        check_chain(gold, rec, lanes=1)


@pytest.mark.parametrize("name", ["cfg1_heights", "normal_n1000", "normal_opts", "beta_bern_n2000", "hier_small", "glm_small"])
@pytest.mark.parametrize("lanes", [4, 64])
def test_oracle_kernel_order_same_decisions(name, lanes):
    gold = golden_io.load(name)
    for rec in gold["chains"][:2]:
        check_chain(gold, rec, lanes=lanes)


@pytest.mark.parametrize("name,lanes", [("hier_small", 64), ("cfg4_full", 64), ("hier_hyper", 64)])      # hier_hyper: 5 groups -- not a power of two
def test_group_local_mode_reproduces_the_reference_decisions(name, lanes):
    """The opt-in group-local evaluation of the hierarchical family (oracle/amwg_oracle.c gl_*: a theta_g proposal is decided on the local
    difference of its group's terms) is not the reference's operation schedule, but it must take the reference's DECISIONS: accept counts,
    adaptation state and uniforms consumed of the seeded reference runs; and its draws must stay within rounding of the reference's."""
    gold = golden_io.load(name)
    case = gold["case"]
    for rec in gold["chains"]:
        spec = model_spec.spec_from_golden(gold, rec)
        o = oracle_lib.OracleChain(spec, case["seed"], rec["chain"], lanes=lanes, group_local=True)
        segs = run_schedule(o, case["schedule"])
        info = o.info()
        assert info["accepts"].tolist() == rec["accepts"]
        assert info["inbounds"].tolist() == rec["inbounds"]
        assert info["batch_count"].tolist() == rec["batch_count"]
        assert o.uniforms() == rec["uniforms"]
        want = np.array(rec["samples"][0]["draws"], dtype=np.float64).reshape(-1, segs[0].shape[1])
        assert np.allclose(segs[0][: want.shape[0]], want, rtol=1e-9, atol=1e-12)
        assert np.allclose(o.state(), rec["final_state"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("model,N,G", [("normal", 300, 0), ("beta_bern", 400, 0), ("hier_normal", 200, 4), ("pois_glm", 120, 0)])
def test_oracle_equals_the_live_reference_on_a_fresh_seed(model, N, G):
    """Beyond the committed goldens: where Node and the reference are at hand (build container: /root/reference; GPU box: oracle/_ref) the
    unmodified reference is run on a seed drawn for this test run and the C restatement must reproduce it bit for bit."""
    import json, os, shutil, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    node = shutil.which("node")
    ref = next((d for d in (os.environ.get("AMWG_REF_DIR"), "/root/reference", os.path.join(root, "oracle", "_ref"))
                if d and os.path.exists(os.path.join(d, "mcmc.js"))), None)
    if node is None or ref is None:
        pytest.skip("node or the reference is not available here")
    seed = int.from_bytes(os.urandom(4), "little") | 1
    case = {"name": "live", "model": model, "N": N, "data_seed": seed ^ 0x2545F491, "store_data": True, "seed": seed, "chains": [0, 11],
            "schedule": [{"op": "burn", "n": 80}, {"op": "sample", "n": 40, "thin": 2}]}
    if G:
        case["G"] = G
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(case, f)
    try:
        p = subprocess.run([node, os.path.join(root, "oracle", "ref_harness.js"), f.name], capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(f.name)
    assert p.returncode == 0, p.stderr
    gold = golden_io._untag(json.loads(p.stdout))
    for rec in gold["chains"]:
        spec = model_spec.spec_from_golden(gold, rec)
        o = oracle_lib.OracleChain(spec, seed, rec["chain"], lanes=1)
        segs = run_schedule(o, case["schedule"])
        want = np.array(rec["samples"][0]["draws"], dtype=np.float64).reshape(-1, segs[0].shape[1])
        assert np.ascontiguousarray(segs[0]).tobytes() == want.tobytes(), "seed %d" % seed
        assert o.info()["accepts"].tolist() == rec["accepts"] and o.uniforms() == rec["uniforms"], "seed %d" % seed
        assert o.state().tolist() == rec["final_state"]


@pytest.mark.parametrize("n_obs,G,seed", [(500, 5, 1), (333, 13, 2), (200, 64, 3), (90, 2, 4)])
def test_group_local_oracle_on_arbitrary_labels_decides_like_the_ordinary_evaluation(n_obs, G, seed):
    """Round 4: the group-local mode for any labels / any G <= 64 (lanes dealt to the groups in aligned power-of-two blocks, gl_layout).  On
    shuffled labels with uneven group sizes the mode must take the decisions of the ordinary evaluation of the same chain (and, the state
    being a function of the decisions only, produce the same draws bit for bit); its log_post agrees to rounding."""
    rng = np.random.default_rng(seed)
    sizes = rng.multinomial(n_obs, rng.dirichlet(np.ones(G) * 2.0))
    g = np.repeat(np.arange(G), sizes).astype(np.int32)
    rng.shuffle(g)
    y = rng.normal(5.0, 3.0, G)[g] + rng.normal(0.0, 2.0, g.size)
    spec = model_spec.build_spec("hier_normal", {"x": y, "g": g, "G": G})
    a = oracle_lib.OracleChain(spec, 5, 3, lanes=1)
    b = oracle_lib.OracleChain(spec, 5, 3, lanes=64, group_local=True)
    a.burn(150)
    b.burn(150)
    da, db = a.sample(50, 1), b.sample(50, 1)
    assert da.tobytes() == db.tobytes()
    assert a.info()["accepts"].tolist() == b.info()["accepts"].tolist() and a.uniforms() == b.uniforms()
    assert abs(a.log_post() - b.log_post()) <= 1e-9 * abs(a.log_post())
