"""The stand-alone stepper classes (reference mcmc.js:1109-1115) without a GPU.

A stepper's log_post takes no arguments and reads the state object it closes over (mcmc.js:428-431).  For every stepper the
scenarios of tests/js/stepper_cases.js create (the reference's own stepper tests, tests/test_mcmc_js.R:52-220, plus shared-state
scenarios), the translator's output -- state object recognised by identity, `function () { return dens(state); }` forwarded to
`dens`, entries of the state no parameter owns laid out as read-only slots -- compiled for the HOST returns bit for bit what the
closure itself returns at 24 states.  tests/js/test_gpu_steppers.js (-m gpu) then checks the classes against the seeded reference.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import user_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")

EXPECTED = ["stepper_real_0", "stepper_int_0", "stepper_multi_real_0", "stepper_multi_int_0", "stepper_binary_0", "stepper_binary_component_0",
            "stepper_amwg_normal_0", "stepper_amwg_complex_0", "stepper_shared_state_0", "stepper_shared_state_1"]


def f64(hexbits):
    return np.frombuffer(bytes.fromhex(hexbits), dtype=">f8")[0]


def test_every_scenario_translates():
    assert user_host.stepper_models() == EXPECTED


@pytest.mark.parametrize("name", EXPECTED)
def test_stepper_closure_equals_itself_on_host(name):
    user_host.stepper_models()
    m = user_host.host_model(name)
    pts = user_host.stepper_states(name)
    assert len(pts) == 24
    finite = 0
    for pt in pts:
        state = [float(f64(h)) for h in pt["state"]]
        got, dv = m.eval(state, 1, derived=True)
        want = f64(pt["lp"])
        assert np.float64(got).tobytes() == np.float64(want).tobytes() or (np.isnan(got) and np.isnan(want)), (name, state, got, want)
        assert [np.float64(a).tobytes() for a in dv] == [np.float64(f64(h)).tobytes() for h in pt["derived"]]
        finite += bool(np.isfinite(want))
    assert finite >= 6, name


def test_shared_state_layout_puts_the_stepped_parameter_first():
    user_host.stepper_models()
    meta = json.load(open(os.path.join(user_host.workdir(), "stepper_shared_state_1.meta.json")))
    assert meta["keys"] == ["sigma", "mu", "shift", "w"] and meta["P"] == 6    # `label` (a string) is not part of the device state


def test_constructor_errors_are_the_references():
    script = r"""
      const mcmc = require('./bayes.js_amd').mcmc;
      const expectThrow = (f, msg) => { try { f(); } catch (e) { if (String(e) !== msg) throw new Error('got ' + e + ' want ' + msg); return; } throw new Error('no throw: ' + msg); };
      const st = { x: 0, y: 1 }, lp = function () { return 0; };
      expectThrow(() => new mcmc.RealMetropolisStepper({ x: { dim: [1] }, y: { dim: [1] } }, st, lp), 'OnedimMetropolisStepper can only handle one parameter.');
      expectThrow(() => new mcmc.IntMetropolisStepper({ x: { dim: [2] } }, st, lp), 'OnedimMetropolisStepper can only handle one one-dimensional parameter.');
      expectThrow(() => new mcmc.MultiRealComponentMetropolisStepper({ x: { dim: [1] }, y: { dim: [1] } }, st, lp), "MultidimComponentMetropolisStepper can't handle more than one parameter.");
      expectThrow(() => new mcmc.BinaryStepper({ x: {}, y: {} }, st, lp), "BinaryStepper can't handle more than one parameter.");
      expectThrow(() => new mcmc.BinaryComponentStepper({ x: {}, y: {} }, st, lp), "BinaryComponentStepper can't handle more than one parameter.");
      expectThrow(() => new mcmc.AmwgStepper({ x: { type: 'complex', dim: [1] } }, st, lp), "AmwgStepper can't handle parameter x with type complex");
      expectThrow(() => new mcmc.RealMetropolisStepper({ z: { dim: [1] } }, st, lp), 'the state has no numeric entry for parameter z');
      for (const k of ['RealMetropolisStepper', 'IntMetropolisStepper', 'MultiRealComponentMetropolisStepper', 'MultiIntComponentMetropolisStepper', 'BinaryStepper', 'BinaryComponentStepper', 'AmwgStepper', 'AmwgSampler', 'runif', 'runif_discrete', 'rnorm', 'param_init_fixed', 'complete_params'])
        if (typeof mcmc[k] !== 'function') throw new Error('missing export ' + k);      // mcmc.js:1103-1117
      console.log('stepper host ok');
    """
    p = subprocess.run([NODE, "-e", script], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "stepper host ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.skipif(not os.path.exists("/root/reference/mcmc.js"), reason="the reference is only present in the build container")
def test_committed_stepper_golden_is_what_the_reference_produces():
    p = subprocess.run([NODE, os.path.join(ROOT, "oracle", "gen_stepper_golden.js")], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout == open(os.path.join(ROOT, "tests", "golden", "steppers.json")).read()


@pytest.mark.gpu
def test_stepper_classes_on_gpu_match_the_seeded_reference():
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_gpu_steppers.js")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "gpu steppers ok" in p.stdout, p.stdout + "\n" + p.stderr
