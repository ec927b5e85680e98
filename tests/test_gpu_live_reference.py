"""-m gpu: the UNMODIFIED reference run LIVE on this box (Node + oracle/ref_harness.js over oracle/_ref, the copy `make -C oracle ref`
leaves beside the oracle -- hashes pinned in oracle/ref.sha256) on seeds and data no committed golden holds, against the HIP path.

The goldens under tests/golden/ were produced in the build container; this test closes the loop on the GPU box itself: a fresh seed per
run (printed; AMWG_LIVE_SEED fixes it), the reference sampler stepping under the Philox `Math.random`, and the GPU sampler through the C
ABI with one lane per chain -- every draw, accept count, adaptation state and uniform count must be bit-identical (mcmc.js:517-553,
985-1039)."""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import amwg_ctypes as A
import golden_io
import model_spec
from gpu_util import run_schedule

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")


def _ref_dir():
    for d in (os.environ.get("AMWG_REF_DIR"), "/root/reference", os.path.join(ROOT, "oracle", "_ref")):
        if d and os.path.exists(os.path.join(d, "mcmc.js")) and os.path.exists(os.path.join(d, "distributions.js")):
            return d
    return None


def _live(case):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(case, f)
        path = f.name
    try:
        p = subprocess.run([NODE, os.path.join(ROOT, "oracle", "ref_harness.js"), path], capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    assert p.returncode == 0, p.stderr
    return golden_io._untag(json.loads(p.stdout))


def _seed():
    return int(os.environ.get("AMWG_LIVE_SEED", "0")) or (int.from_bytes(os.urandom(4), "little") | 1)


needs_ref = pytest.mark.skipif(NODE is None or _ref_dir() is None, reason="node or the reference copy (oracle/_ref, made by `make -C oracle ref`) is missing")


@needs_ref
def test_reference_copy_is_the_pinned_unmodified_reference():
    import hashlib
    want = dict(line.split()[::-1] for line in open(os.path.join(ROOT, "oracle", "ref.sha256")) if line.strip())
    for name, digest in want.items():
        assert hashlib.sha256(open(os.path.join(_ref_dir(), name), "rb").read()).hexdigest() == digest, name


@needs_ref
@pytest.mark.parametrize("model,N,G,burn,n", [("normal", 700, 0, 260, 120), ("beta_bern", 900, 0, 200, 100), ("hier_normal", 330, 5, 120, 60), ("pois_glm", 240, 0, 90, 40)])
def test_live_reference_equals_gpu_on_a_fresh_seed(model, N, G, burn, n):
    seed = _seed()
    case = {"name": "live", "model": model, "N": N, "data_seed": seed ^ 0x5bd1e995, "store_data": True, "seed": seed,
            "chains": [0, (seed % 60000) + 1], "schedule": [{"op": "burn", "n": burn}, {"op": "sample", "n": n, "thin": 3}]}
    if G:
        case["G"] = G
    print("live reference: model %s seed %d" % (model, seed))
    gold = _live(case)
    for rec in gold["chains"]:
        spec = model_spec.spec_from_golden(gold, rec)
        s = A.Sampler(spec, chains=2, seed=seed, chain_offset=rec["chain"], lanes_per_chain=1)
        segs = run_schedule(s, case["schedule"])
        for got, want in zip(segs, rec["samples"]):
            assert got.shape[0] == want["kept"]
            w = np.array(want["draws"], dtype=np.float64).reshape(-1, got.shape[1])
            assert np.ascontiguousarray(got[:, :, 0]).tobytes() == w.tobytes(), "seed %d" % seed
        info, d = s.info(), s.diag()
        assert s.state()[:, 0].tolist() == rec["final_state"]
        assert info["accepts"][:, 0].tolist() == rec["accepts"]
        assert info["inbounds"][:, 0].tolist() == rec["inbounds"]
        assert info["prop_log_scale"][:, 0].tolist() == rec["prop_log_scale"]
        assert info["batch_count"][:, 0].tolist() == rec["batch_count"]
        assert int(d["uniforms"][0]) == rec["uniforms"]
        assert float(d["log_post"][0]) == rec["log_post"]
        s.close()
