#!/usr/bin/env python3
"""bench.py -- the AMWG hot path on N MI355X, measured the way BASELINE.json asks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1] -- Normal(mu, sigma) model, 1e4 synthetic
observations (SURVEY.md §8d recipe), 65 536 chains PER GPU (weak scaling; chain ids are global, so
rank r runs chains [r*65536, (r+1)*65536) of one logical job).  A "step" is one Sampler.step()
(mcmc.js:985-997) of every chain = P = 2 parameter updates per chain.  W untimed steps, then EXACTLY
K steps timed between barrier + synchronize pairs; the K steps are one sample() call (chunked into
kernel launches of --steps-per-launch steps) that also records every `--thin`-th draw into HBM,
followed for N > 1 by the RCCL gather of the recorded draws to rank 0 (the "gather at sample collection" of north_star).  Inputs are resident in
HBM before the timed region; the D2H copy of draws is outside it (see DESIGN.md for the PCIe-inclusive rate).

value = (chains on all GPUs) * K * P / seconds of the MEDIAN timed region  [param-updates/s].  The K-step region (barrier +
    synchronize on both sides, max over ranks) is repeated until >= 1 s has been timed (at least 3, at most 400 regions) and the
    median region is reported, so that the driver's short `--steps 20` run reports the steady state instead of one launch
    on a chip that has not clocked up yet; `timing` carries every region's figure incl. the first.
roofline: the binding roof of this path is fp64 VALU issue (78.6 TFLOP/s = 3.93e13 lane-operations/s), NOT HBM: the data vector
    is staged once per launch into LDS and re-read from there.  roofline.frac = algorithmic fp64 lane-operations (8 per
    observation: sub, mul, the 4-operation correctly rounded quotient, sub, add) / second / peak.  The SURVEY.md section 8(d)
    "effective bandwidth" (80 024 algorithmic bytes per update / launch time, vs the 8 TB/s HBM peak) is reported beside it as
    roofline.effective_hbm with lds_resident: true -- it exceeds 1 by design.  roofline.traffic = measured HBM bytes per launch
    (rocprofv3 PMC, profiles/), traffic_ratio = traffic / algorithmic bytes.
cpu_baseline = the UNMODIFIED reference (bench/ref_cpu.js: node + $AMWG_REF_DIR | /root/reference | oracle/_ref, one thread, median of 5
    >= 1 s repeats) when Node and the reference are present ("kind": "reference"); on a box without the reference (the GPU
    box) the C port oracle/amwg_oracle.c on one thread ("kind": "port", "reference_unavailable": true).
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]

import numpy as np  # noqa: E402

N_OBS = 10_000
CHAINS_PER_GPU = 65_536
SEED = 20260925
DATA_SEED = 20260925
B_ALG_PER_UPDATE = N_OBS * 8 + 8 * 2 + 8     # 80 024 B: data once + state read + draw write
HBM_PEAK_GBPS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK = 78.6e12 / 2                 # lane-FMA/s: 256 CU * 4 SIMD * 16 lanes/clk * 2.4 GHz
FP64_FLOPS_PEAK = 78.6e12                    # the same peak in SURVEY.md section 8(d)'s unit (an FMA = 2 flop)
# SURVEY.md section 8(d), "Algorithmic flops (secondary)": ld.norm's term as the reference writes it is ~5 fp64 flops per observation INCLUDING
# one division (sub, mul, div, sub, add).  gfx950 has no fp64 divide instruction: the correctly rounded quotient is 4 issued operations
# (amwg_div.h: mul + 3 fma; IEEE '/' expands to 11), which is why the issue-based figure counts 8 per observation.  Both are printed.
SURVEY_FLOPS_PER_OBS = {"normal": 5, "hier_normal": 5}


def roofline_units(fam, lane_ops_per_s, ops_per_obs):
    """frac_issue: issued-operation view (what the VALU has to execute, the roof the kernel runs against); frac_survey_flops: the same run in
    SURVEY.md section 8(d)'s flop count (a division = one flop) against the 78.6 TFLOP/s datasheet figure (an FMA = two flops)."""
    out = {"frac_issue": lane_ops_per_s / FP64_VALU_PEAK}
    f = SURVEY_FLOPS_PER_OBS.get(fam)
    if f is not None:
        out["frac_survey_flops"] = lane_ops_per_s / ops_per_obs * f / FP64_FLOPS_PEAK
        out["survey_flops_note"] = ("SURVEY.md section 8(d) counts ~%d flops per observation incl. ONE division; against 78.6 TFLOP/s (FMA = 2 flops) the same run is frac_survey_flops. "
                                    "frac (= frac_issue) counts the %d operations the VALU must issue per observation (the division alone is 4: no fp64 divide on gfx950) "
                                    "against the 3.93e13 lane-operations/s the SIMDs can issue" % (f, ops_per_obs))
    return out


HIER_SWEEP_PASSES = 1      # amwg_sweep_kernel with certified decisions (round 5): ONE pass per step -- the sweep's sums of squares about the proposed means; neither
HIER_SWEEP_OPS = 2         # mu's nor sigma's update reads the data (a lane's sum of squares depends on its mean only) -- of 2 operations per observation (sub, fma)


def sweep_lane_ops(updates_per_s, P, n_obs):
    """fp64 lane-operations/s of the hierarchical family's sweep kernel: a Sampler.step of P updates makes HIER_SWEEP_PASSES pass(es) over the data (not P), each
    n_obs x HIER_SWEEP_OPS; the stepper's own arithmetic (proposals, butterflies, accept tests) is not counted as algorithmic work."""
    return updates_per_s / P * HIER_SWEEP_PASSES * n_obs * HIER_SWEEP_OPS


HIER_SWEEP_OPS_NOTE = ("2 = sub, fma: the sweep kernel's one pass per step forms every lane's sum of squares about the proposed mean of its group; the accept tests of the whole step "
                       "(32 theta, mu, sigma) are decided from those sums with a rigorous bound on their distance from the reference's term-by-term expression (8 operations per "
                       "observation), which is evaluated when a uniform falls inside the bound and once per launch and chain; every draw bit-identical to the reference's")


def sweep_roofline_units(lane_ops_per_s):
    return {"frac_issue": lane_ops_per_s / FP64_VALU_PEAK, "frac_survey_flops": lane_ops_per_s / HIER_SWEEP_OPS * 3 / FP64_FLOPS_PEAK,
            "survey_flops_note": "the certified pass in flops (a sub and an fma per observation = 3, an FMA = 2) against 78.6 TFLOP/s"}


SWEEP_NOTE = ("roofline of the kernel that produces `value` (%s): since it decides from certified sums (csrc/amwg_models.h HierNormalModel::sweep_approx / log_post_approx) a step of "
              "%d updates reads the data ONCE -- the sweep's sums of squares about the proposed means, %d observations x 2 fp64 operations; mu's and sigma's updates need no pass "
              "(a lane's sum of squares depends on its mean only).  Against that arithmetic the kernel is far from the fp64 roof BY CONSTRUCTION: what it issues is the stepper "
              "(~2 500 vector + ~1 800 scalar instructions per step-wave, profiles/), two wavefronts per SIMD, latency bound (valu_pipe_busy 0.44).  The kernel that passes over "
              "all the data in EVERY update (options.full_evaluation = 1, the reference's expression) runs at %.3g param-updates/s, %.3f of the roof in its own unit (8 operations "
              "per observation, one pass per update)")


def normal_spec():
    import synth
    data = synth.normal(N_OBS, DATA_SEED)
    opt = {"prop_log_scale": 0.0, "batch_size": 50, "max_adaptation": 0.33, "initial_adaptation": 1.0,
           "target_accept_rate": 0.44, "is_adapting": True}                     # mcmc.js:500-505
    inf = float("inf")
    params = [  # complete_params() of {mu:{type:"real"}, sigma:{type:"real", lower:0}} (README.md:22-24)
        {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": -inf, "upper": inf, "init": [0.5]},
        {"type": "real", "len": 1, "top": 1, "multidim": 0, "lower": 0.0, "upper": inf, "init": [0.5]}]
    return {"model": "normal", "n_obs": N_OBS, "data": data, "params": params, "P": 2, "init": [0.5, 0.5],
            "comp_opts": [dict(opt), dict(opt)], "G": 0, "K": 0}


# The other BASELINE.json configs (parity-test cases; measurable with --workload for DESIGN.md's table, never the default):
#   name: (family, n_obs, chains per GPU, algorithmic bytes per update, fp64 lane-ops per observation, label)
OTHER_WORKLOADS = {
    # cfg3: with one lane per chain the two-valued sequential sum is fast-forwarded over binades (csrc/amwg_models.h two_valued_sum,
    # bit-identical to the term-by-term pass), so an update no longer streams the data: the "algorithmic bytes" figure is nominal
    "cfg3": ("beta_bern", 100_000, 262_144, 100_000 * 1 + 8 * 1 + 8, 1, "BASELINE.json configs[2]: Beta-Bernoulli, 1e5 binary obs, 262144 chains per GPU (exact fast-forward of the two-valued sum)"),
    "cfg4": ("hier_normal", 10_000, 2_048, 10_000 * 9 + 8 * 34 + 8, 8, "BASELINE.json configs[3]: hierarchical Normal (34 components), 1e4 obs, 2048 chains per GPU (16384 over 8)"),
    # the one number the reference publishes (README.md:252): Normal model, 1000 data points, 20 000 draws
    # "~0.5 s" = 8.0e4 param-updates/s on the author's machine -- ONE chain, so this measures single-chain latency
    "readme": ("normal", 1_000, 1, 1_000 * 8 + 8 * 2 + 8, 8, "README.md:252 claim: Normal(mu,sigma), 1000 obs, ONE chain (run with --steps 20000)"),
    "cfg5": ("pois_glm", 50_000, 8_192, 50_000 * (7 * 8 + 8 + 8) + 8 * 9 + 8, 86, "BASELINE.json configs[4]: Poisson GLM + int change point, 5e4 obs, 8192 chains per GPU (65536 over 8)"),
}


# what the fp64 lane-operations per observation of each family are (the roofline's unit of arithmetic)
CERTIFIED_OPS_PER_OBS = 2      # the certified pass of the Normal family with one lane per chain: sub, fma (sum of (x - mu)^2; csrc/amwg_pass.h norm_sq_pass_uniform)
CERTIFIED_NOTE = ("2 = sub, fma: by default the accept test of the Normal family (one lane per chain) is decided from prior + n c - sum (x - mu)^2 / den with a rigorous bound on its "
                  "distance from the reference's term-by-term expression (csrc/amwg_kernel.h 'certified decisions'); every update still passes over all the observations; the "
                  "expression itself (8 operations per observation) is evaluated when a uniform falls inside the bound (~1e-7 of the updates) and once per launch and chain. "
                  "Every draw is bit-identical to the reference's (parity block).  `full_evaluation` beside this: the kernel that evaluates the expression in every update")
CERTIFIED_GLM_OPS_PER_OBS = 27
CERTIFIED_GLM_NOTE = ("27 = the certified pass of the Poisson family (16 lanes per chain, the four chains of a wavefront sharing every row they read): linear predictor as one "
                      "product + six fmas (7), change point (1), exp_bounded (17: k = round(x / ln2), two fused reduction steps, a degree-11 interpolating polynomial by Horner's rule "
                      "in fmas, conversion of k, ldexp), sum eta y (1 fma), sum lambda (1).  The logarithm of the exponential the reference takes is NOT formed: the value is used "
                      "with a rigorous bound on its distance from the expression's (csrc/amwg_models.h PoisGlmModel::log_post_approx); the expression itself (86 operations per "
                      "observation) is evaluated when a uniform falls inside the bound and once per launch and chain.  `full_evaluation` beside this: the expression in every update")


def certified_kind(fam, lanes, full_evaluation=False):
    """which certified pass the default geometry runs (None: the reference's expression in every update)"""
    if full_evaluation:
        return None
    if fam == "normal" and lanes == 1:
        return "normal"
    if fam == "pois_glm" and lanes == 16:
        return "pois_glm"
    return None


OPS_NOTE = {
    "normal": "8 = sub, mul, 4-operation correctly rounded quotient (amwg_div.h: mul, fma, fma, fma), sub, add; IEEE '/' would be 17",
    "hier_normal": "8 = sub, mul, 4-operation correctly rounded quotient, sub, add (the gather of theta[g_i] is an LDS read, not arithmetic)",
    "beta_bern": "1 = the fp64 add of the term-by-term pass (the observation selects WHICH register is added, on the scalar unit)",
    "pois_glm": "86 = the fp64 operations the expression needs once the operations exp and log have in common are formed once: 13 linear "
                "predictor (7 mul + 6 add, no contraction), 1 change point (add; the comparison is an integer one), 32 V8 exp (3 k: mul, add, trunc; "
                "4 hi/lo/r; 1 r*r; 10 polynomial; 10 r*c/(2-c) incl. the 8-operation quotient; 3 reassembly; 1 int conversion of k), 36 V8 log of that value (its "
                "argument split and its two k*ln2 products are exp's own: 2 f and 2+f; 8 quotient; 14 polynomials; 12 both tails and the final sum), "
                "4 density + accumulate.  ISSUED per observation (disassembly, utilisation, not the roofline's unit): ~105 incl. integer/select/address "
                "work (138 until round 3's second half).  Practical ceiling of the issue rate: an fp64 instruction every ~4.45 cycles with two waves "
                "per SIMD and 16.5 for v_rcp_f64 (tools/ubench/valu_rates.hip)",
}


def other_spec(name, exp):
    import model_spec
    fam, n_obs = OTHER_WORKLOADS[name][0], OTHER_WORKLOADS[name][1]
    return model_spec.build_spec(fam, model_spec.make_data(fam, n_obs, DATA_SEED, G=32, exp=exp))


def kernel_id_of(version):
    m = re.search(r"kernels ([0-9a-f]{12})", version or "")
    return m.group(1) if m else None


def kernel_base_name(name):
    """'void amwg::amwg_sweep_kernel<amwg::HierNormalModel, 512>(amwg::StepArgs)' (rocprofv3) and 'amwg_sweep_kernel<HierNormalModel,512> with ...'
    (this file) -> 'amwg_sweep_kernel<HierNormalModel,512>'"""
    n = (name or "").replace("amwg::", "").replace(" ", "")
    n = n[4:] if n.startswith("void") else n
    m = re.match(r"[\w]+(<[^>]*>)?", n)
    return m.group(0) if m else n


def measured_traffic(chains, steps_per_launch, workload="cfg2", lanes=None, group_local=False, kernel_id=None, kernel=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_summary.json, written by
    tools/profile.sh + tools/summarize_profile.py for this same command).  PMC counters need rocprofv3 around the process, so the figure
    cannot be taken inside this run; what ties it to the run is the KERNEL ID (tools/build_id.py: a hash of the device sources + compiler
    flags, carried by amwg_version() and stored with every profile): a profile of other kernel sources is refused.
    -> (bytes, file, algorithmic bytes of that launch, why-not)"""
    import glob
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), key=os.path.getmtime, reverse=True):
        try:
            p = json.load(open(f))
        except (OSError, ValueError):
            continue
        if p.get("workload", "cfg2") != workload or ("--group-local" in p.get("command", "")) != bool(group_local):
            continue
        # the profile must be of the same kernel instantiation: the kernel the roofline figure is about, by name (cfg4: the full-evaluation step kernel, not
        # the sweep kernel that produces `value`); a name without its workgroup class (cfg3's side measurement) is matched on Model and lanes
        want, have = kernel_base_name(kernel), kernel_base_name(p.get("kernel", ""))
        if kernel is not None and re.search(r",\d+,\d+>$|^amwg_(sweep|gl)_kernel|^amwg_user_step", want):
            if want != have:
                continue
        elif lanes is not None and not re.search(r",\s*%d(,\s*\d+)?>" % lanes, p.get("kernel", "")):
            continue
        if p.get("chains") == chains and p.get("steps_per_launch") == steps_per_launch and p.get("hbm_traffic_bytes_per_launch"):
            if kernel_id is not None and p.get("kernel_id") != kernel_id:
                stale = stale or "%s is of kernels %s, this library is kernels %s: refused" % (os.path.relpath(f, ROOT), p.get("kernel_id", "(no id: profiled before round 4)"), kernel_id)
                continue
            return p["hbm_traffic_bytes_per_launch"], os.path.relpath(f, ROOT), p.get("algorithmic_bytes_per_launch"), None
    return None, None, None, stale or "no profile of this workload / geometry under profiles/"


VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.0      # wave64 vector instructions per second: 1024 SIMDs, 16 lanes per clock each (MI355X_MICROARCH.md) -- the same rate as the fp64 peak, counted in instructions


def profiled_valu_per_update(workload, kernel, kernel_id):
    """Vector instructions one parameter update costs a wavefront's LANE in the kernel `kernel`, from the committed rocprofv3 PMC pass of the same kernel sources
    (profiles/r*_<workload>_summary.json: SQ_INSTS_VALU of the profile's adapted launches / a launch's updates).  What a PASS-FREE update is priced with (cfg3's exact fast-forward: the
    stepper -- Philox, Leva's rnorm, two logarithms, the bisection over the binades -- is all there is): achieved = value x this, against the chip's vector issue rate.
    -> (wave64 instructions per 64 updates, file, why-not)"""
    import glob
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), key=os.path.getmtime, reverse=True):
        try:
            p = json.load(open(f))
        except (OSError, ValueError):
            continue
        if p.get("workload") != workload or kernel_base_name(p.get("kernel", "")) != kernel_base_name(kernel):
            continue
        # (the MINIMUM over the profiled launches: the launches of a sampler whose proposal scales are still adapting reject more often inside rnorm and issue more;
        # the rate this is multiplied with is measured on adapted launches)
        v = ((p.get("pmc_per_launch") or {}).get("SQ_INSTS_VALU") or {}).get("min")
        upd = (p.get("chains") or 0) * (p.get("steps_per_launch") or 0) * (p.get("components") or 1)
        if not v or not upd:
            continue
        if kernel_id is not None and p.get("kernel_id") != kernel_id:
            stale = stale or "%s is of kernels %s, this library is kernels %s: refused" % (os.path.relpath(f, ROOT), p.get("kernel_id"), kernel_id)
            continue
        return v / (upd / 64.0), os.path.relpath(f, ROOT), None
    return None, None, stale or "no profile of this kernel under profiles/"


def end_to_end_js(chains, n_obs):
    """SURVEY.md section 8(d) asks for kernel-only AND end-to-end: the same job through the JavaScript host (bench/js_e2e.js: require, constructor incl.
    translation + hiprtc or the on-disk code-object cache, burn(1000), sample(1000) INCLUDING the copy of every draw to the host and into the
    arrays sample() returns), plus the reference's own use -- one chain, constructed twice.  None when Node or the addon is missing."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None or not os.path.exists(os.path.join(ROOT, "bayes.js_amd", "csrc", "amwg_napi.node")):
        return None
    try:
        p = subprocess.run([node, "--max-old-space-size=8192", os.path.join(ROOT, "bench", "js_e2e.js"), "--chains", str(chains), "--n-obs", str(n_obs)],
                           capture_output=True, text=True, timeout=300)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        r = json.loads(line)
    except (subprocess.SubprocessError, ValueError, IndexError, OSError) as e:
        return {"error": repr(e)}
    m = r["many_chains"]
    out = {"what": "node bench/js_e2e.js: new mcmc.AmwgSampler(README closure, %d obs, {chains: %d}) + burn(%d) + sample(%d) incl. the copy-out of every draw + close()"
                   % (m["n_obs"], m["chains"], m["burn"], m["sample"]),
           "updates_per_s": m["updates_per_s"], "updates_per_s_excl_constructor": m["updates_per_s_excl_ctor"], "unit": "param-updates/s",
           "ctor_ms": m["ctor_ms"], "burn_ms": m["burn_ms"], "sample_ms_incl_copy_out": m["sample_ms"], "total_ms": m["total_ms"], "gb_copied": m["gb_copied"],
           "lanes_per_chain": (m.get("launch") or [{}])[0].get("lanes_per_chain"), "node": r["node"], "require_ms": r["require_ms"], "code_cache": r.get("code_cache")}
    t = r.get("many_chains_translated")
    if t:      # the same job with a closure the family recogniser does not know: translated + hiprtc; certified decisions from the translator (round 6)
        out["translated_many_chains"] = {"what": "the README model with its priors swapped (translate.js -> hiprtc), same data, chains and schedule",
                                         "kernel": ((t.get("launch") or [{}])[0]).get("kernel"), "updates_per_s_excl_constructor": t["updates_per_s_excl_ctor"],
                                         "ctor_ms": t["ctor_ms"], "burn_ms": t["burn_ms"], "sample_ms_incl_copy_out": t["sample_ms"],
                                         "kernel_only_updates_per_s": (2.0 * t["chains"] * t["sample"] / (((t.get("launch") or [{}])[0]).get("kernel_ms", 0.0) * 1e-3)) if ((t.get("launch") or [{}])[0]).get("kernel_ms") else None}
    # a closure that has to be translated and compiled: a process with an empty code-object cache, then a second process that finds it on disk
    try:
        import tempfile
        with tempfile.TemporaryDirectory() as cache:
            env = dict(os.environ, AMWG_CACHE_DIR=cache)
            runs = []
            for _ in range(2):
                q = subprocess.run([node, os.path.join(ROOT, "bench", "js_e2e.js"), "--translated-only", "1"], capture_output=True, text=True, timeout=120, env=env)
                runs.append(json.loads([ln for ln in q.stdout.splitlines() if ln.startswith("{")][-1]))
        out["translated_closure"] = {"what": "README model with the priors swapped (not a recognised family): translate.js -> hiprtc; ONE chain, ten heights; two processes sharing one cache directory",
                                     "first_process": {"ctor_ms": runs[0]["translated"]["ctor_ms"], "code_cache": runs[0]["code_cache"]},
                                     "second_process": {"ctor_ms": runs[1]["translated"]["ctor_ms"], "code_cache": runs[1]["code_cache"]},
                                     "burn1000_sample5000_ms": runs[1]["translated"]["burn_sample_ms"]}
    except (subprocess.SubprocessError, ValueError, IndexError, OSError, KeyError) as e:
        out["translated_closure"] = {"error": repr(e)}
    if "single_chain_first" in r:
        a, b = r["single_chain_first"], r["single_chain_again"]
        out["single_chain"] = {"what": "README.md:18-43 as is: ONE chain, ten heights, burn(1000) + sample(5000); constructed twice in one process",
                               "first": {"ctor_ms": a["ctor_ms"], "burn_ms": a["burn_ms"], "sample_ms": a["sample_ms"], "total_ms": a["total_ms"], "updates_per_s": a["updates_per_s"]},
                               "again": {"ctor_ms": b["ctor_ms"], "burn_ms": b["burn_ms"], "sample_ms": b["sample_ms"], "total_ms": b["total_ms"], "updates_per_s": b["updates_per_s"]}}
    return out


def reference_dir():
    """Where the unmodified reference can be loaded from: $AMWG_REF_DIR, /root/reference (build container), or oracle/_ref -- the
    copy `make -C oracle ref` (run by __graft_entry__.build()) leaves beside the oracle; git-ignored, it travels to the GPU box."""
    for d in (os.environ.get("AMWG_REF_DIR"), "/root/reference", os.path.join(ROOT, "oracle", "_ref")):
        if d and os.path.exists(os.path.join(d, "mcmc.js")) and os.path.exists(os.path.join(d, "distributions.js")):
            return d
    return None


def cpu_baseline_reference(workload):
    """The unmodified reference on one host core (bench/ref_cpu.js); None where Node or the reference is missing."""
    import shutil
    import subprocess
    node = shutil.which("node")
    ref = reference_dir()
    if node is None or ref is None:
        return None
    try:
        out = subprocess.run([node, os.path.join(ROOT, "bench", "ref_cpu.js"), "--workload", workload, "--ref", ref],
                             capture_output=True, text=True, timeout=120, check=True).stdout.strip().splitlines()[-1]
        r = json.loads(out)
    except (subprocess.SubprocessError, ValueError, IndexError, OSError):
        return None
    if not r.get("available"):
        return None
    return {"value": r["value"], "unit": "param-updates/s", "cores": 1, "kind": "reference", "unmodified_sha256_ok": r.get("unmodified"),
            "sample": "unmodified %s/mcmc.js under Node %s, same model+data (N=%d), 1 chain, median of 5 burn(%d) repeats (%.2f s each) after "
                      "%d warm-up steps; the reference evaluates log_post twice per update (mcmc.js:524-526)"
                      % (ref, r["node"], r["n_obs"], r["steps_per_repeat"], r["median_s"], r["warmup_steps"]),
            "repeats_s": r["repeats_s"]}


def cpu_baseline_port(spec, budget_s=10.0):
    """The oracle (a C port of the reference algorithm), single thread, same data, 1 chain."""
    import oracle_lib
    ch = oracle_lib.OracleChain(spec, SEED, 0, lanes=1)
    t0 = time.perf_counter()
    ch.burn(5)
    rate = 5 / (time.perf_counter() - t0)
    warm = int(min(1000, max(5, rate * 3.0)))   # adapted, steady state (1000 steps where a step is cheap enough)
    ch.burn(warm)
    t0 = time.perf_counter()
    ch.burn(max(5, min(500, int(rate))))
    rate = max(5, min(500, int(rate))) / (time.perf_counter() - t0)
    n = max(5, int(rate * budget_s))
    t0 = time.perf_counter()
    ch.burn(n)
    dt = time.perf_counter() - t0
    return {"value": n * spec["P"] / dt, "unit": "param-updates/s", "cores": 1, "kind": "port", "reference_unavailable": True,
            "sample": "oracle/amwg_oracle.c (C restatement of mcmc.js:517-553 + distributions.js, one log_post pass per update), same model+data "
                      "(N=%d), 1 chain, %d steps after >= %d burn-in (%.1f s).  The unmodified JS reference is not on this box (no "
                      "/root/reference, AMWG_REF_DIR unset); on the build container bench/ref_cpu.js measures it at 3.6e4 for cfg2 "
                      "(profiles/r02_ref_cpu.json), i.e. ~15 %% below this port" % (spec["n_obs"], n, warm, dt)}


def cpu_baseline_all_cores(spec, single_rate, budget_s=2.0, max_threads=32):
    """Optional stronger baseline (ours, not the reference's): the same C oracle, one independent
    chain per thread on up to 32 threads (ctypes releases the GIL), ~2 s of work per thread."""
    import concurrent.futures
    import oracle_lib
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cores = max(1, min(usable, max_threads))
    n = max(5, int(single_rate / spec["P"] * budget_s))
    chains = [oracle_lib.OracleChain(spec, SEED, c, lanes=1) for c in range(cores)]
    for ch in chains:
        ch.burn(2)
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda ch: ch.burn(n), chains))
    dt = time.perf_counter() - t0
    return {"value": cores * n * spec["P"] / dt, "unit": "param-updates/s", "cores": cores, "kind": "port",
            "sample": "oracle/amwg_oracle.c, %d independent chains on %d threads (of %d usable), %d steps each (%.1f s); ours, not the reference" % (cores, cores, usable, n, dt)}


GOLDEN_OF = {"cfg2": "cfg2_full", "cfg3": "cfg3_full", "cfg4": "cfg4_full", "cfg5": "cfg5_full"}


def flip_rate_record(kernel_id=None):
    """The decision-parity campaign (tools/flip_rate.py, committed under profiles/): how many chains of a seeded job ever decide differently at 64
    lanes / group-local than with one lane per chain (the reference's summation order), over 1e9+ decisions.  Like roofline.traffic, the figure is
    REFUSED when the campaign ran other kernels than the library that is being timed (kernel id of amwg_version())."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flip_rate.json")))      # (by name: rNN sorts by round; checkouts do not keep mtimes)
    if not fs:
        return None
    try:
        r = json.load(open(fs[-1]))
    except (OSError, ValueError):
        return None
    if kernel_id is not None and kernel_id_of(r.get("version")) != kernel_id:
        return {"source": os.path.relpath(fs[-1], ROOT), "refused": "the campaign ran kernels %s, this library is kernels %s: re-run tools/flip_rate.py" % (kernel_id_of(r.get("version")), kernel_id),
                "library_version_of_campaign": r.get("version")}
    return {"source": os.path.relpath(fs[-1], ROOT), "decisions_total": r["decisions_total"], "first_flips_total": r["first_flips_total"],
            "flips_per_1e9": r["flips_per_1e9"], "upper_95_per_1e9": r["upper_95_per_1e9"], "library_version_of_campaign": r.get("version"),
            "reference_order": r.get("reference_order"),
            "per_run": [{"workload": q["workload"], "geometry": q["geometry"], "chains": q["chains"], "steps": q["steps"], "decisions": q["decisions"],
                         "chains_differing": q["chains_differing"], "lp_abs_diff_max": q["lp_abs_diff_max"], "expected_flips_bound": q["expected_flips_bound"]} for q in r["runs"]],
            "note": "one lane per chain IS the reference's order (bit-identical draws), and so are the multi-lane defaults of cfg4 / cfg5 (reference_order: decided against the "
                    "expression in that order -- zero chains may differ); in the other multi-lane kernels (flips_per_1e9) a decision can differ only when the accept uniform "
                    "falls inside the ~1e-12-relative sliver between the two summation orders' exp(delta): counted here, chain by chain"}


def golden_schedule_check(s, gold, locals_and_records, lanes):
    """Runs the golden case's schedule ON THE GIVEN (full-size) SAMPLER -- device-resident draws -- and compares the listed local chains with
    the seeded run of the unmodified reference stored in tests/golden/<case>.json: accept counts, in-bounds counts, adaptation state and
    uniforms consumed always (the reference's decisions); where the sampler decides in the reference's summation order (amwg_summation_order() == 1: one
    lane per chain, and the certified multi-lane kernels of cfg4 / cfg5 since round 5) also every stored draw, the running sums over all kept draws, the
    final state and the cached log_post, bit for bit.  -> dict of booleans."""
    import torch
    case = gold["case"]
    P, C = s.P, s.C
    seg_draws = []
    for seg in case["schedule"]:
        if seg["op"] == "burn":
            s.burn(seg["n"])
        elif seg["op"] == "sample":
            thin = seg.get("thin", 1)
            rows = -(-seg["n"] // thin)
            d = torch.empty((rows, P, C), dtype=torch.float64, device="cuda")
            s.sample_device(seg["n"], thin, d.data_ptr(), d.numel() * 8)
            s.sync()
            seg_draws.append(d)
    info, diag, state = s.info(), s.diag(), s.state()
    ok = {"accept_counts_identical": True, "uniforms_consumed_identical": True, "adaptation_state_identical": True}
    lanes = s.launch_info().get("summation_order", lanes)      # (1: the reference's own order whatever the lane count)
    if lanes == 1:
        ok.update({"draws_bit_identical": True, "running_sums_bit_identical": True, "final_state_bit_identical": True, "log_post_bit_identical": True})
    for local, rec in locals_and_records:
        ok["accept_counts_identical"] &= info["accepts"][:, local].tolist() == rec["accepts"] and info["inbounds"][:, local].tolist() == rec["inbounds"]
        ok["uniforms_consumed_identical"] &= int(diag["uniforms"][local]) == rec["uniforms"]
        ok["adaptation_state_identical"] &= (info["batch_count"][:, local].tolist() == rec["batch_count"] and
                                              info["acceptance_count"][:, local].tolist() == rec["acceptance_count"])
        if lanes == 1:
            ok["adaptation_state_identical"] &= info["prop_log_scale"][:, local].tolist() == rec["prop_log_scale"]
            ok["final_state_bit_identical"] &= state[:, local].tolist() == rec["final_state"]
            ok["log_post_bit_identical"] &= float(diag["log_post"][local]) == rec["log_post"]
            for d, want in zip(seg_draws, rec["samples"]):
                col = d[:, :, local].cpu().numpy()                      # (rows, P) of ONE chain
                w = np.array(want["draws"], dtype=np.float64).reshape(-1, P)
                ok["draws_bit_identical"] &= col.shape[0] == want["kept"] and np.ascontiguousarray(col[: w.shape[0]]).tobytes() == w.tobytes()
                tot = np.zeros(P)
                for t in range(col.shape[0]):
                    tot = tot + col[t]
                ok["running_sums_bit_identical"] &= tot.tolist() == want["sum"]
    del seg_draws
    return {k: bool(v) for k, v in ok.items()}


def timed_geometry_parity(A, spec, workload, chains, make_sampler, lanes):
    """BASELINE.md section 3, item 6 / round-2 review: the bit-for-bit proof taken OUT OF THE FULL-SIZE SAMPLER that is timed -- same chain
    count, workgroup geometry and lane count as the timed launches.  `make_sampler(chain_offset)` builds it; the golden's chain ids are
    global, so a chain id beyond this GPU's shard is looked up in the shard that holds it (cfg4: chain 16383 = the last chain of the 8th
    shard of 2048).  Returns (report, the sampler of shard 0 -- advanced by the golden schedule, reused as the timed one)."""
    import golden_io
    gold = golden_io.load(GOLDEN_OF[workload])
    first, report, checked, order = None, None, [], lanes
    by_shard = {}
    for rec in gold["chains"]:
        by_shard.setdefault(rec["chain"] // chains, []).append(rec)
    for shard in sorted(by_shard):
        smp = make_sampler(shard * chains)
        order = smp.launch_info().get("summation_order", lanes)
        r = golden_schedule_check(smp, gold, [(rec["chain"] - shard * chains, rec) for rec in by_shard[shard]], lanes)
        checked += [rec["chain"] for rec in by_shard[shard]]
        report = r if report is None else {k: report[k] and r[k] for k in r}
        if shard == 0:
            first = smp
        else:
            smp.close()
    sched = gold["case"]["schedule"]
    report.update({"golden": "tests/golden/%s.json (seeded run of the unmodified reference)" % GOLDEN_OF[workload], "chains_checked": checked,
                   "schedule": sched, "from_timed_sampler": True, "chains_in_sampler": chains, "lanes_per_chain": lanes,
                   "summation_order": order, "reference_order": order == 1,
                   "note": ("one lane per chain: every draw of every chain is the reference's, bit for bit" if lanes == 1 else
                            "%d lanes per chain, decisions certified against the expression in the REFERENCE's order (amwg_summation_order() == 1): every draw, the final "
                            "state and the cached log_post of the checked chains are the reference's, bit for bit" % lanes if order == 1 else
                            "%d lanes per chain: the sum over observations is formed in lane order, so doubles are compared with the oracle in the same "
                            "order by the test suite; here: every accept decision, adaptation step and uniform count equals the reference's" % lanes)})
    return report, first


def measure_sufficient_statistics(A, spec, chains, device):
    """cfg2 with options.sufficient_statistics = 1 (opt-in; include/amwg.h): the cheaper value of log_post the accept test is decided from comes from the data's two
    sufficient statistics -- sum (x - mu)^2 = SS + n (xbar - mu)^2 -- instead of a pass, with the same bound and the same fallback to the reference's expression: the
    same draws bit for bit (tests/test_gpu_parity.py), no O(n) work per update.  Reported here, NEVER as `value`: the headline's kernel passes over all observations in
    every update.  HIP events around adapted burn launches, as for the other configs; the first chain's draws are compared with the default sampler's."""
    mk = lambda suff: A.Sampler(spec, chains=chains, seed=SEED, device=device, steps_per_launch=100, sufficient_statistics=suff)
    a, b = mk(1), mk(0)
    same = True
    for q in (a, b):
        q.burn(150)
    da, db = a.sample(40, 1), b.sample(40, 1)
    same = da[:, :, :64].tobytes() == db[:, :, :64].tobytes() and a.state().tobytes() == b.state().tobytes()
    b.close()
    a.burn(800)
    a.burn(500)
    li = a.launch_info()
    value = chains * 500 * spec["P"] / (li["kernel_ms"] * 1e-3)
    a.close()
    return {"value": value, "unit": "param-updates/s", "kernel": li["kernel"], "opt_in": "options.sufficient_statistics = 1", "draws_equal_the_default_samplers": bool(same),
            "note": "the certified value from SS + n (xbar - mu)^2: no pass over the data; what is left is the stepper (Philox, rnorm, exp, accept, adaptation)"}


def measure_other_config(A, name, device, group_local=0):
    """A short driver-visible measurement of one of the other BASELINE.json configs at its per-GPU size: golden check out of the full-size
    sampler, then HIP-event time of adapted launches.  -> dict for the bench line's `other_configs`."""
    fam, n_obs, chains, b_alg, ops_per_obs, label = OTHER_WORKLOADS[name]
    spec = other_spec(name, A.lib().amwg_exp)
    P = spec["P"]
    mk = lambda off: A.Sampler(spec, chains=chains, seed=SEED, chain_offset=off, device=device, steps_per_launch=100, group_local=group_local)
    probe = mk(0)
    lanes = probe.launch_info()["lanes_per_chain"]
    probe.close()
    t0 = time.perf_counter()
    parity, s = timed_geometry_parity(A, spec, name, chains, mk, lanes)
    # adapted steady state: the golden schedule has already stepped the chains; a little more, then the timed launches
    per_step = chains * P
    warm = {"cfg3": 300, "cfg4": 400, "cfg5": 20}[name]
    timed = {"cfg3": 300, "cfg4": 600, "cfg5": 20}[name]
    s.burn(warm)
    s.burn(timed)
    li = s.launch_info()
    kernel_s = li["kernel_ms"] * 1e-3
    value = per_step * timed / kernel_s
    out = {"workload": label, "value": value, "unit": "param-updates/s", "chains": chains, "n_obs": n_obs, "components": P, "steps_timed": timed,
           "timing": "HIP events around the %d launches of one burn(%d) call after the golden schedule + %d more steps" % (li["n_launches"], timed, warm),
           "lanes_per_chain": li["lanes_per_chain"], "block_threads": li["block_threads"], "grid_blocks": li["grid_blocks"], "parity": parity}
    roof_updates_per_s, kernel, note = value, "amwg_step_kernel<%s,%d>" % ({"beta_bern": "BetaBernModel", "hier_normal": "HierNormalModel", "pois_glm": "PoisGlmModel"}[fam], lanes), None
    if name == "cfg3":
        t = A.Sampler(spec, chains=chains, seed=SEED, device=device, lanes_per_chain=1, steps_per_launch=20, exact_division=1)
        t.burn(40)
        t.burn(20)
        roof_updates_per_s = chains * 20 * P / (t.launch_info()["kernel_ms"] * 1e-3)
        t.close()
        # `value` and `frac` describe the SAME kernel (round-5 review, item 5): the default's update is pass-free -- the exact fast-forward of the two-valued sum
        # (csrc/amwg_twoval.h) -- so what bounds it is vector ISSUE: the stepper's instructions per update (from the committed PMC profile of these kernel sources)
        # x the measured update rate, against the chip's issue rate.  The term-by-term pass (exact_division = 1) is reported beside it.
        out["term_by_term_value"] = roof_updates_per_s
        out["term_by_term_frac"] = roof_updates_per_s * n_obs * ops_per_obs / FP64_VALU_PEAK
        kernel = li.get("kernel") or kernel
        vpu, vfile, why = profiled_valu_per_update("cfg3", kernel, kernel_id_of(A.lib().amwg_version().decode()))
        out["roofline"] = {"bound": "valu_issue", "achieved": (value / 64.0) * vpu if vpu else None, "peak": VALU_ISSUE_PEAK, "unit": "wave64 vector instructions/s",
                           "frac": ((value / 64.0) * vpu / VALU_ISSUE_PEAK) if vpu else None, "kernel": kernel, "valu_per_64_updates": vpu, "profile": vfile, "profile_refused": why,
                           "note": "pass-free update (exact fast-forward over ~log2 N binades): Philox + Leva's rnorm + two logarithms + the bisection; the roof is vector issue, "
                                   "priced with the instructions per update of the committed rocprofv3 PMC pass of the same kernel sources.  Term by term (exact_division = 1, one fp64 "
                                   "add per observation): %.3g param-updates/s, %.2f of the fp64 rate" % (roof_updates_per_s, out["term_by_term_frac"]),
                           "effective_hbm_gbps": value * b_alg / 1e9}
        out["seconds"] = time.perf_counter() - t0
        s.close()
        return out
    sweep_kernel = False
    if name == "cfg4" and not group_local:
        # by default only the lanes whose sum an update can have changed are re-formed (csrc/amwg_models.h lane_sum_rows: bit-identical to evaluating
        # everything, like the cached log_post of the current state), so `value` no longer streams the data once per update: the roofline figure is
        # the kernel that does (options.full_evaluation = 1), measured on the side
        t = A.Sampler(spec, chains=chains, seed=SEED, device=device, steps_per_launch=100, full_evaluation=1)
        t.burn(200)
        t.burn(300)
        roof_updates_per_s = chains * 300 * P / (t.launch_info()["kernel_ms"] * 1e-3)
        t.close()
        out["full_evaluation_value"] = roof_updates_per_s
        out["full_evaluation_frac"] = roof_updates_per_s * n_obs * ops_per_obs / FP64_VALU_PEAK
        out["value_kernel"] = li.get("kernel")
        if str(li.get("kernel", "")).startswith("amwg_sweep_kernel"):
            # `value` and `frac` describe the SAME kernel (round-4 review): the sweep kernel against the arithmetic of its three passes
            kernel = li["kernel"]
            note = SWEEP_NOTE % (kernel, P, n_obs, roof_updates_per_s, out["full_evaluation_frac"])
            roof_updates_per_s = None
            sweep_kernel = True
        else:
            kernel += " with options.full_evaluation = 1"
            note = "roofline = every evaluation passes over all the data (full_evaluation = 1), %.3g param-updates/s" % roof_updates_per_s
    ops_note = OPS_NOTE[fam]
    if name == "cfg5" and certified_kind(fam, li["lanes_per_chain"]):
        # the default runs the certified pass (27 operations per observation, four chains sharing a row): `value` and `frac` describe that kernel; the
        # kernel that evaluates the reference's expression in every update is measured beside it
        t = A.Sampler(spec, chains=chains, seed=SEED, device=device, lanes_per_chain=li["lanes_per_chain"], steps_per_launch=10, full_evaluation=1)
        t.burn(10)
        t.burn(10)
        fv = chains * 10 * P / (t.launch_info()["kernel_ms"] * 1e-3)
        out["full_evaluation_value"], out["full_evaluation_frac"] = fv, fv * n_obs * ops_per_obs / FP64_VALU_PEAK
        t.close()
        ops_per_obs, ops_note = CERTIFIED_GLM_OPS_PER_OBS, CERTIFIED_GLM_NOTE
        kernel = li.get("kernel", kernel) + " (certified decisions)"
    lane_ops = sweep_lane_ops(value, P, n_obs) if roof_updates_per_s is None else roof_updates_per_s * n_obs * ops_per_obs
    if group_local:
        kernel = li.get("kernel") or kernel      # (amwg_gl_kernel<HierGlModel,BT>: what a profiler lists -- round-5 review: the line named the plain step kernel here)
        # group-local evaluation: a step of the P = G + 2 updates makes TWO passes over the data (the sweep over theta and the sigma update)
        # instead of P; the fp64 work per update is what those two passes do, not one pass per update
        lane_ops = roof_updates_per_s * (2.0 / P) * n_obs * ops_per_obs
        label += " -- GROUP-LOCAL evaluation (amwg_options::group_local, opt-in: not the reference's operation schedule; decisions identical on every golden and over the flip-rate campaign, see parity.flip_rate)"
        out["workload"] = label
        note = ("group-local: the %d proposals for theta of a step are evaluated in one pass (every lane with the proposed mean of its own group) and decided on "
                "their local differences, mu needs no pass, sigma one: 2 passes per step instead of %d (mcmc.js:524-526 makes 2 per update).  roofline = "
                "the arithmetic of those two passes; the rest of a step is the stepper's serial logic" % (P - 2, P))
    out["roofline"] = {"bound": "fp64_valu", "achieved": lane_ops, "peak": FP64_VALU_PEAK, "frac": lane_ops / FP64_VALU_PEAK, "unit": "fp64 lane-operations/s",
                       "lane_ops_per_obs": ops_per_obs, "lane_ops_note": ops_note, "kernel": kernel, "note": note,
                       "effective_hbm_gbps": value * b_alg / 1e9}
    if sweep_kernel:
        out["roofline"].update(lane_ops_per_obs=HIER_SWEEP_OPS, lane_ops_note=HIER_SWEEP_OPS_NOTE, **sweep_roofline_units(lane_ops))
    else:
        out["roofline"].update(roofline_units(fam, lane_ops, ops_per_obs))
    if lanes > 1 and not group_local:
        out["reference_order"] = reference_order_price(A, name, spec, device)
        if "value" in out["reference_order"]:
            out["reference_order_value"] = out["reference_order"]["value"]
    if name in TRANSLATED_TWIN and not group_local:
        out["translated_closure"] = measure_translated_twin(A, name, chains, device, warm, timed)
    out["seconds"] = time.perf_counter() - t0
    s.close()
    return out


# The same config written as a PLAIN JavaScript closure that the family recogniser does not know (tests/js/user_models.js bench_hier / bench_glm): translated by
# bayes.js_amd/translate.js, compiled with hiprtc, certified decisions from the translator's own plans (docs/CERTIFIED.md: row plan / Poisson tail).  Reported beside
# the hand-written family's `value`, never as it.  Needs node (the translator is JavaScript); absent -> {"skipped": ...}.
TRANSLATED_TWIN = {"cfg4": "bench_hier", "cfg5": "bench_glm"}


def measure_translated_twin(A, name, chains, device, warm, timed):
    import shutil
    if shutil.which("node") is None:
        return {"skipped": "node is not installed: the closure translator is JavaScript"}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import decision_parity
        spec = decision_parity.user_spec_of(TRANSLATED_TWIN[name])
        t = A.Sampler(spec, chains=chains, seed=SEED, device=device, steps_per_launch=100)
        t.burn(warm + (150 if name == "cfg4" else 40))      # (the family's sampler above has also run its golden schedule before its timed launches)
        t.burn(timed)
        li = t.launch_info()
        t.close()
        return {"closure": "tests/js/user_models.js " + TRANSLATED_TWIN[name], "value": chains * spec["P"] * timed / (li["kernel_ms"] * 1e-3), "unit": "param-updates/s",
                "kernel": li.get("kernel"), "lanes_per_chain": li["lanes_per_chain"], "block_threads": li["block_threads"], "summation_order": li.get("summation_order"),
                "timing": "HIP events around the launches of one burn(%d) call" % timed}
    except Exception as e:      # (a reported side measurement: never the reason a bench line is missing)
        return {"error": "%s: %s" % (type(e).__name__, e)}


# What strict identity costs (round-4 review, item 7; mcmc.js:527-528): north_star asks for bit-identical accept counts.  With ONE lane per chain the
# sum over observations is the reference's `lp += term`, so every draw and every decision is the reference's.  Until round 5 the default geometry of cfg4 /
# cfg5 summed in its own lane order (3 of 3.7e10 decisions differed); now those kernels decide against the expression in the reference's order
# (amwg_summation_order() == 1: strict identity at no price), and the figure below -- the same config forced to lanes_per_chain = 1, with the chain count raised
# to the whole 8-GPU job so that the one-lane launch has lanes to fill the chip with (stated) -- is what strict identity cost before, and still costs the
# kernels that have no certified path (translated closures at > 1 lane, options.full_evaluation != 0).
REFERENCE_ORDER_RUN = {"cfg4": (16_384, 20, 40), "cfg5": (65_536, 2, 3)}      # chains, warm-up steps, timed steps


def reference_order_price(A, name, spec, device):
    chains, warm, timed = REFERENCE_ORDER_RUN[name]
    try:
        t = A.Sampler(spec, chains=chains, seed=SEED, device=device, lanes_per_chain=1, steps_per_launch=timed)
        t.burn(warm)
        t.burn(timed)
        li = t.launch_info()
        t.close()
    except Exception as e:
        return {"error": repr(e)}
    return {"value": chains * timed * spec["P"] / (li["kernel_ms"] * 1e-3), "unit": "param-updates/s", "lanes_per_chain": 1, "chains": chains, "block_threads": li["block_threads"],
            "grid_blocks": li["grid_blocks"], "kernel": li.get("kernel"), "steps_timed": timed,
            "note": "options.lanes_per_chain = 1: the reference's summation order, every draw and accept count bit-identical to mcmc.js; %d chains (the whole 8-GPU job on this one "
                    "GPU: a one-lane launch needs that many to occupy the chip), HIP events around %d steps after %d" % (chains, timed, warm)}


def main_inproc(args):
    """`--inproc`: the PRODUCT's own multi-device path, which is what a JavaScript user gets from `options.devices`: ONE process, one sampler
    per device holding a contiguous shard of the global chain ids, every device driven from this thread with amwg_burn_async /
    amwg_sample_async + amwg_sync, and the posterior summary of the sharded job formed inside the library (amwg_group_moments: per-device
    reductions + RCCL all-reduce over the shards' devices).  No torch.distributed.  Weak scaling by default (the workload's chains per GPU on
    every device); `--strong` keeps the job's TOTAL chain count (cfg4: 16 384, cfg5: 65 536, cfg2: 65 536, cfg3: 262 144) and splits it."""
    import torch
    N = args.gpus
    have = torch.cuda.device_count()
    base = {"metric": "posterior draws/sec (= param-updates/sec)", "unit": "param-updates/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "inproc": True}
    if have < N:
        emit(dict(base, value=None, ms_per_step=None, config={"workload": args.workload},
                  note="not measured: --inproc --gpus %d needs %d visible devices, this box has %d" % (N, N, have)))
        return
    import amwg_ctypes as A
    if args.workload == "cfg2":
        spec, per_gpu, label, n_obs, ops_per_obs = normal_spec(), CHAINS_PER_GPU, "BASELINE.json configs[1]: Normal(mu,sigma) AMWG, 1e4 synthetic obs", N_OBS, 8
        total_job = CHAINS_PER_GPU
    else:
        spec = other_spec(args.workload, A.lib().amwg_exp)
        _, n_obs, per_gpu, _, ops_per_obs, label = OTHER_WORKLOADS[args.workload]
        total_job = {"cfg3": 262_144, "cfg4": 16_384, "cfg5": 65_536, "readme": 1}[args.workload]
    if args.chains_per_gpu != CHAINS_PER_GPU:
        per_gpu = args.chains_per_gpu
    total = total_job if args.strong else per_gpu * N
    P, K, W, thin = spec["P"], args.steps, args.warmup, max(1, args.thin)
    shards, off = [], 0
    for r in range(N):
        cnt = total // N + (1 if r < total % N else 0)
        shards.append(A.Sampler(spec, chains=cnt, seed=SEED, chain_offset=off, device=r, lanes_per_chain=args.lanes, block_threads=args.block,
                                steps_per_launch=args.steps_per_launch, group_local=int(args.group_local)))
        off += cnt
    for s in shards:
        s.burn_async(W)
    for s in shards:
        s.sync()

    def region():
        t0 = time.perf_counter()
        for s in shards:
            s.sample_async(K, thin)            # draws stay in each device's HBM
        for s in shards:
            s.sync()
        A.group_gather_draws(shards, root=0, to_host=False)      # the gather at sample collection: every shard's block to device 0 (amwg_group_gather_draws: grouped ncclSend / ncclRecv)
        mean, sd = A.group_moments(shards)     # per-device sums, RCCL all-reduce (inside the library)
        return time.perf_counter() - t0, mean, sd
    regs = [region()]
    n_regions = 1 if args.single_region else int(min(400, max(3, np.ceil(args.min_seconds / max(regs[0][0], 1e-6)))))
    while len(regs) < n_regions:
        regs.append(region())
    order = sorted(range(len(regs)), key=lambda i: regs[i][0])
    dt, mean, sd = regs[order[len(order) // 2]]
    value = total * K * P / dt
    kernel_ms = max(s.launch_info()["kernel_ms"] for s in shards)
    li = shards[0].launch_info()
    # how the draws reach the HOST: what the Node front-end does (every device copies its own block: N PCIe links in parallel) against
    # gather-to-one-device-then-one-copy (north_star's wording); both once, outside the timed regions
    collection = {}
    t0 = time.perf_counter()
    for s in shards:
        s.fetch_draws()
    collection["per_device_copy_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    A.group_gather_draws(shards, root=0, to_host=True)
    collection["gather_then_one_copy_ms"] = (time.perf_counter() - t0) * 1e3
    collection["bytes"] = sum(s._pending * s.PR * s.C * 8 for s in shards)
    comm = A.group_comm_info(shards)
    sweep = str(li.get("kernel", "")).startswith("amwg_sweep_kernel")
    lane_ops = sweep_lane_ops(value, P, n_obs) if sweep else value * n_obs * ops_per_obs * ((2.0 / P) if args.group_local else 1.0)
    out = dict(base, value=value, ms_per_step=dt * 1e3 / K,
               config={"workload": label + (" -- GROUP-LOCAL evaluation" if args.group_local else ""), "n_obs": n_obs, "chains_total": total,
                       "chains_per_gpu": [s.C for s in shards], "components": P, "thin": thin, "lanes_per_chain": li["lanes_per_chain"],
                       "block_threads": li["block_threads"], "steps_per_launch": args.steps_per_launch,
                       "path": "one process, one sampler per device (amwg_sample_async x N + amwg_sync), the recorded draws gathered to device 0 by amwg_group_gather_draws, summaries by amwg_group_moments (RCCL inside the library)",
                       "rccl_ranks_seen": comm["rccl_ranks_seen"], "devices": comm["devices"], "collection_to_host": collection},
               timing={"regions": len(regs), "reported": "median region (wall clock around sample_async x N + sync x N + group_gather_draws + group_moments)",
                       "region_ms": [r[0] * 1e3 for r in regs][:64], "slowest_device_kernel_ms_last_region": kernel_ms},
               roofline={"bound": "fp64_valu", "achieved": lane_ops, "peak": FP64_VALU_PEAK * N, "frac": lane_ops / (FP64_VALU_PEAK * N),
                         "unit": "fp64 lane-operations/s", "lane_ops_per_obs": HIER_SWEEP_OPS if sweep else ops_per_obs,
                         "lane_ops_note": HIER_SWEEP_OPS_NOTE if sweep else OPS_NOTE[spec["model"]], "kernel": li.get("kernel"),
                         "note": ("the sweep kernel: one certified 2-operation pass over the data per step of %d updates" % P) if sweep else None},
               posterior={"mean": mean.tolist()[:8], "sd": sd.tolist()[:8], "note": "amwg_group_moments over the recorded draws of all shards (last region)"})
    emit(out)
    for s in shards:
        s.close()


_REAL_STDOUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries loaded into this process write there too -- RCCL prints a five-line version banner
    through C stdio when a communicator is created, which lands AFTER the line when stdout is a pipe -- so file descriptor 1 is pointed at
    stderr for the whole run and emit() writes the line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
LINE_LIMIT = 4096      # bytes of the stdout line (round-4 review: a 19 KB line was not parsed by the driver; tests assert < 8192)


def _num(x, sig=7):
    """floats of the stdout line at 7 significant digits (the detail file keeps every bit)"""
    if isinstance(x, float) and x == x and abs(x) != float("inf"):
        return float("%.*g" % (sig, x))
    return x


def _pick(d, *keys):
    return {k: _num(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, detail_path=None):
    """The ONE line of stdout: the contract's fields and nothing that grows with the run.  Everything else `out` holds (region lists, notes, flip-rate runs,
    the JavaScript end-to-end figures, the other configs' parity reports) goes to the detail file and to stderr."""
    line = {k: _num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, "workload", "n_obs", "chains_per_gpu", "chains_total", "components", "lanes_per_chain", "block_threads", "steps_per_launch", "thin", "rccl_ranks_seen")
    if isinstance(line["config"].get("chains_per_gpu"), list) and len(line["config"]["chains_per_gpu"]) > 8:
        line["config"]["chains_per_gpu"] = line["config"]["chains_per_gpu"][:8]
    for k in ("inproc", "note", "kernel_only_value", "chains_equiv"):
        if out.get(k) is not None:
            line[k] = _num(out[k])
    if isinstance(out.get("full_evaluation"), dict):      # the kernel that evaluates the reference's expression in every update, beside the default
        line["full_evaluation"] = _pick(out["full_evaluation"], "value", "frac")
    if isinstance(out.get("term_by_term"), dict):         # cfg3: the term-by-term pass beside the fast-forward
        line["term_by_term"] = _pick(out["term_by_term"], "value", "frac")
    r = out.get("roofline")
    if r:
        line["roofline"] = _pick(r, "bound", "achieved", "peak", "unit", "frac", "frac_of_measured_peak", "kernel", "launch_ms", "traffic", "traffic_ratio")
        line["roofline"].setdefault("traffic", None)
        line["roofline"].setdefault("frac", None)
        if r.get("traffic") is None and r.get("traffic_refused"):
            line["roofline"]["traffic_refused"] = str(r["traffic_refused"])[:120]
        if r.get("effective_hbm"):
            line["roofline"]["effective_hbm"] = _pick(r["effective_hbm"], "achieved", "peak", "unit", "frac", "lds_resident")
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = _pick(c, "value", "unit", "cores", "kind", "reference_unavailable")
        line["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:160]
    p = out.get("parity")
    if p:
        line["parity"] = _pick(p, "accept_counts_identical", "uniforms_consumed_identical", "adaptation_state_identical", "draws_bit_identical", "final_state_bit_identical", "lanes_per_chain", "reference_order")
        fr = p.get("flip_rate")
        if fr:
            line["parity"]["flip_rate"] = _pick(fr, "flips_per_1e9", "upper_95_per_1e9", "decisions_total", "first_flips_total", "refused")
            ro = fr.get("reference_order")
            if ro:      # the kernels that decide in the reference's order (cfg4 / cfg5 defaults): chains that differ from the one-lane run, of how many decisions
                line["parity"]["flip_rate"]["reference_order"] = _pick(ro, "decisions_total", "chains_differing", "log_post_differs")
    oc = out.get("other_configs")
    if oc:
        line["other_configs"] = {}
        for name, o in oc.items():
            if "error" in o:
                line["other_configs"][name] = {"error": str(o["error"])[:100]}
                continue
            q = _pick(o, "value", "chains", "lanes_per_chain", "reference_order_value", "full_evaluation_value", "term_by_term_value", "term_by_term_frac")
            rr = o.get("roofline") or {}
            q.update(_pick(rr, "frac"))
            if rr.get("kernel"):
                q["kernel"] = kernel_base_name(rr["kernel"])
            pp = o.get("parity") or {}
            q["parity_ok"] = bool(pp) and all(v for k, v in pp.items() if k.endswith("_identical"))
            if pp.get("summation_order") is not None:
                q["summation_order"] = pp["summation_order"]      # (1: the reference's own order -- draws, final state and log_post are among the *_identical above)
            tc = o.get("translated_closure")
            if isinstance(tc, dict) and "value" in tc:      # the same config as a plain closure through translate.js + hiprtc (reported beside the family's value)
                q["translated_closure_value"] = _num(tc["value"])
            line["other_configs"][name] = q
    ss = out.get("sufficient_statistics")
    if isinstance(ss, dict):
        line["sufficient_statistics"] = _pick(ss, "value", "opt_in", "draws_equal_the_default_samplers", "error")
    e = out.get("end_to_end_js")
    if isinstance(e, dict) and "updates_per_s" in e:
        line["end_to_end_js"] = _pick(e, "updates_per_s", "total_ms", "updates_per_s_excl_constructor", "ctor_ms", "sample_ms_incl_copy_out")
        tm = e.get("translated_many_chains")
        if tm:
            line["end_to_end_js"]["translated"] = _pick(tm, "kernel", "updates_per_s_excl_constructor", "kernel_only_updates_per_s")
        sc = (e.get("single_chain") or {}).get("again")
        if sc:
            line["end_to_end_js"]["single_chain_warm_ms"] = _num(sc.get("total_ms"))
    lib = out.get("library")
    if lib:
        line["library"] = {"kernels": kernel_id_of(lib.get("version")), "built_from_this_tree": lib.get("built_from_this_tree")}
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    text = json.dumps(line, separators=(",", ":"))
    # a line over the limit sheds its optional parts, least important first, rather than going out unparseable
    for drop in ("end_to_end_js", "library", "kernel_only_value", "chains_equiv", "other_configs", "parity"):
        if len(text) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        line["dropped_for_size"] = line.get("dropped_for_size", []) + [drop]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, "bench line is %d bytes" % len(text)
    return text


def emit(obj, detail_path=None):
    """stdout: compact_line(obj), ONE line <= 4 KB.  The whole record: `detail_path` (default bench_detail.json beside this file) and stderr."""
    detail_path = detail_path or os.environ.get("AMWG_BENCH_DETAIL") or DETAIL_FILE
    full = json.dumps(obj, indent=1)
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        sys.stderr.write("bench.py: cannot write %s: %r\n" % (detail_path, e))
        detail_path = None
    sys.stderr.write("".join("| " + ln + "\n" for ln in full.splitlines()))      # (prefixed: no line of stderr can be taken for the JSON line)
    sys.stderr.flush()
    line = (compact_line(obj, detail_path) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--thin", type=int, default=1, help="record every thin-th draw inside the timed region (default 1: SURVEY.md section 8(d) recipe, `sample 1000 (thin 1)`)")
    ap.add_argument("--chains-per-gpu", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--steps-per-launch", type=int, default=100,
                    help="steps fused into one kernel launch; warm-up and timed steps use the same launch size so the "
                         "per-launch time bench.py reports is comparable with rocprofv3's per-kernel average")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--group-local", action="store_true", help="cfg4 only: the opt-in group-local evaluation (amwg_options::group_local)")
    ap.add_argument("--inproc", action="store_true", help="the product's own multi-device path: one process, one sampler per device, amwg_group_moments (see main_inproc)")
    ap.add_argument("--strong", action="store_true", help="keep the job's TOTAL chain count (cfg4: 16 384, cfg5: 65 536, cfg2: 65 536, cfg3: 262 144) and split it over the GPUs; the default for cfg4 / cfg5, which is how BASELINE.json states them")
    ap.add_argument("--weak", action="store_true", help="the workload's chains per GPU on every GPU (the default for cfg2 / cfg3: 65 536 / 262 144 per GPU)")
    ap.add_argument("--full-evaluation", action="store_true", help="cfg4: options.full_evaluation = 1 for the timed sampler too (every evaluation passes over all the data; profiling the roofline kernel)")
    ap.add_argument("--torch-gather", action="store_true", help="N > 1: gather the draws with torch.distributed instead of the library's own communicator (amwg_comm_*)")
    ap.add_argument("--no-parity", action="store_true", help="profiling runs only: skip the golden schedule on the timed sampler (its launches record every draw)")
    ap.add_argument("--no-js", action="store_true", help="skip the end-to-end run through the JavaScript host (bench/js_e2e.js)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short measurements of cfg3 / cfg4 / cfg5 appended to the default line")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the timed K-step region until this much time is on the clock (median reported)")
    ap.add_argument("--single-region", action="store_true", help="time the K-step region once (profiling runs)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "readme"],
                    help="cfg2 (default) is the bench line; the others measure the remaining BASELINE.json configs")
    args = ap.parse_args()
    if not args.strong and not args.weak and args.workload in ("cfg4", "cfg5"):
        args.strong = True      # BASELINE.json: "16 384 chains sharded over 8 MI355X", "65 536 chains over 8 MI355X"

    if args.inproc:
        return main_inproc(args)
    import torch
    import amwg_ctypes as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # `python bench.py --gpus N` without torch.distributed.run: never leave without a line.  With N devices visible this is the PRODUCT's own
            # multi-device path (one process, one sampler per device, amwg_group_gather_draws: main_inproc); otherwise the "not measured" line, rc 0.
            sys.stderr.write("bench.py: --gpus %d without torch.distributed.run (WORLD_SIZE unset): the one-process multi-device path (--inproc)\n" % args.gpus)
            return main_inproc(args)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available() and world == 1:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # (N ranks started by the launcher on a box with fewer than N devices -- none included: the "not measured" line below, rc 0, like the launcher-less call)
    # Development switches (not used by the driver): AMWG_BENCH_BACKEND=gloo + AMWG_BENCH_ONE_DEVICE=1 run the
    # N-rank code path on a box with a single GPU (ranks share cuda:0, gather goes through host memory).
    backend = os.environ.get("AMWG_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("AMWG_BENCH_ONE_DEVICE") == "1" else local_rank
    if os.environ.get("AMWG_BENCH_ONE_DEVICE") != "1" and world > torch.cuda.device_count():
        if rank == 0:
            emit({"metric": "posterior draws/sec (= param-updates/sec)", "value": None, "unit": "param-updates/s", "n_gpus": world, "steps": args.steps,
                  "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
                  "dtype": "f64", "data": "synthetic", "config": {"workload": args.workload},
                  "note": "not measured: --gpus %d needs %d visible devices, this box has %d" % (world, world, torch.cuda.device_count())})
        return
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":      # RCCL over xGMI
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    from shard import chain_shard, gather_draws
    b_alg, ops_per_obs, n_obs = B_ALG_PER_UPDATE, 8, N_OBS
    label = "BASELINE.json configs[1]: Normal(mu,sigma) AMWG, 1e4 synthetic obs, 65536 chains per GPU"
    if args.workload == "cfg2":
        spec = normal_spec()
    else:
        spec = other_spec(args.workload, A.lib().amwg_exp)
        _, n_obs, default_chains, b_alg, ops_per_obs, label = OTHER_WORKLOADS[args.workload]
        if args.chains_per_gpu == CHAINS_PER_GPU:
            args.chains_per_gpu = default_chains
    chains = args.chains_per_gpu
    total_job = {"cfg2": CHAINS_PER_GPU, "cfg3": 262_144, "cfg4": 16_384, "cfg5": 65_536, "readme": 1}[args.workload]
    if args.strong:      # the job's total chain count, split: rank r gets a contiguous shard (sizes differ by at most one)
        offset, chains = chain_shard(rank, world, total_job)
        total_chains_job = total_job
    else:
        offset, _ = chain_shard(rank, world, chains * world)
        total_chains_job = chains * world
    mk = lambda off: A.Sampler(spec, chains=chains, seed=SEED, chain_offset=off, device=dev_index, group_local=int(args.group_local), full_evaluation=int(args.full_evaluation),
                               lanes_per_chain=args.lanes, block_threads=args.block, steps_per_launch=args.steps_per_launch)
    parity = None
    if world == 1 and args.workload in GOLDEN_OF and not args.no_parity:
        # the proof that this build reproduces the reference, taken from the sampler that is timed below (its first steps ARE the golden schedule)
        probe = mk(0)
        timed_lanes = probe.launch_info()["lanes_per_chain"]
        probe.close()
        parity, s = timed_geometry_parity(A, spec, args.workload, chains, mk, timed_lanes)
    else:
        s = mk(offset)
    P, K, W, thin = spec["P"], args.steps, args.warmup, max(1, args.thin)
    rows = -(-K // thin)
    draws = torch.empty((rows, P, chains), dtype=torch.float64, device="cuda")
    # The gather at sample collection is the LIBRARY's (amwg_comm_*: a communicator over the ranks from a shared id, grouped ncclSend / ncclRecv of
    # every rank's block to rank 0) -- what a one-process-per-device host of the product calls; torch.distributed only carries the 128-byte id,
    # the barriers and the max-over-ranks of the timing.  --torch-gather (or a failure to build the communicator, recorded in the line) times
    # torch's gather of the same blocks instead.
    comm, comm_error, comm_info = None, None, None
    if backend == "nccl" and not args.torch_gather:
        def exchange(ident):
            if dist is None:
                return ident
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        # the headline must not be lost to a communicator that cannot be built -- or whose construction never returns (ncclCommInitRank is
        # collective and blocking: it runs on a helper thread with a deadline; the reason is printed either way)
        import threading
        made = {}

        def make():
            try:
                made["comm"] = A.Comm(world, rank, dev_index, exchange)
                made["info"] = made["comm"].info()
            except Exception as e:
                made["error"] = repr(e)
        th = threading.Thread(target=make, daemon=True)
        th.start()
        th.join(float(os.environ.get("AMWG_BENCH_COMM_DEADLINE_S", "180")))
        if th.is_alive():
            comm_error = "amwg_comm_create did not return within its deadline"
        elif "error" in made:
            comm_error = made["error"]
        else:
            comm, comm_info = made["comm"], made["info"]
        if dist is not None:      # all ranks or none
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0 and comm is not None:
                comm.close()
                comm, comm_error = None, "another rank could not build the communicator"
    gather_elems = rows * P * total_chains_job
    gathered_lib = torch.empty(gather_elems, dtype=torch.float64, device="cuda") if (comm is not None and rank == 0) else None
    shard_sizes = [chain_shard(r, world, total_chains_job)[1] for r in range(world)] if args.strong else [chains] * world
    gathered = [torch.empty((rows, P, shard_sizes[r]), dtype=draws.dtype, device=coll_dev) for r in range(world)] if (comm is None and world > 1 and rank == 0) else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if W > 0:
        s.burn(W)

    def timed_region():
        """EXACTLY K steps between barrier + synchronize pairs; -> (wall seconds, HIP-event kernel ms), max over ranks"""
        barrier()
        t0 = time.perf_counter()
        s.sample_device(K, thin, draws.data_ptr(), draws.numel() * 8)
        s.sync()
        if comm is not None:
            comm.gather_draws(s, 0, gathered_lib.data_ptr() if rank == 0 else 0, gather_elems * 8)
        elif dist is not None:
            gather_draws(dist, draws if coll_dev == "cuda" else draws.cpu(), gathered, rank, equal_sizes=len(set(shard_sizes)) == 1)
        barrier()
        dt = time.perf_counter() - t0
        kms = s.launch_info()["kernel_ms"]
        if dist is not None:
            t = torch.tensor([dt, kms], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, kms = float(t[0]), float(t[1])
        return dt, kms

    # the K-step region, repeated until >= 1 s is on the clock (3..400 regions; every rank derives the same count from the
    # max-over-ranks time of the first region), median region reported
    regions = [timed_region()]
    n_regions = int(min(400, max(3, np.ceil(args.min_seconds / max(regions[0][0], 1e-6)))))
    if args.single_region:
        n_regions = 1
    while len(regions) < n_regions:
        regions.append(timed_region())
    order = sorted(range(len(regions)), key=lambda r: regions[r][0])
    dt, kernel_ms = regions[order[len(order) // 2]]
    li = s.launch_info()
    # posterior moments over the recorded draws of ALL ranks: per-rank sums + all-reduce (RCCL for N > 1; bayes.js_amd/shard.py,
    # the one-process-per-GPU twin of the library's amwg_group_moments) -- outside the timed regions
    from shard import pooled_moments
    pooled = pooled_moments(dist, draws if (dist is None or coll_dev == "cuda") else draws.cpu())

    lib_moments = comm.moments(s) if comm is not None else None      # collective: the library's all-reduce twin of pooled_moments
    rank_devices = [dev_index]
    if dist is not None:
        box = [None] * world
        dist.all_gather_object(box, {"rank": rank, "device": dev_index, "chains": chains, "rccl": comm_info})
        rank_devices = box
    if rank == 0:
        total_chains = total_chains_job
        value = total_chains * K * P / dt
        launches = max(1, li["n_launches"])
        launch_s = kernel_ms * 1e-3 / launches
        updates_per_launch = chains * (K / launches) * P
        eff_gbps = updates_per_launch * b_alg / launch_s / 1e9
        mean, sd = (s.moments() if dist is None else (pooled[0].cpu().numpy(), pooled[1].cpu().numpy()))
        pm = pooled[0].cpu().numpy()
        assert dist is not None or np.allclose(pm, mean, rtol=1e-10, atol=0), "library moments differ from the pooled restatement"
        if lib_moments is not None:
            assert np.allclose(lib_moments[0], pm, rtol=1e-9, atol=1e-300), "amwg_comm_moments differs from the pooled restatement"
            if gathered_lib is not None:      # rank 0's own block is where its rank says (blocks back to back in rank order)
                own = gathered_lib[: rows * P * chains].view(rows, P, chains)
                assert torch.equal(own, draws), "amwg_comm_gather_draws: rank 0's block is not where it belongs"
        measured_peak = A.fp64_peak(dev_index)      # register-only fma kernel: what the chip sustains under fp64 load
        x = spec["data"]["x"]
        version = A.lib().amwg_version().decode()
        kname = {"normal": "NormalModel", "beta_bern": "BetaBernModel", "hier_normal": "HierNormalModel", "pois_glm": "PoisGlmModel"}[spec["model"]]
        bt_class = 256 if li["block_threads"] <= 256 else (512 if li["block_threads"] <= 512 else 1024)
        kernel = li.get("kernel") or "amwg_step_kernel<%s,%d,%d>" % (kname, li["lanes_per_chain"], bt_class)      # (amwg_kernel_name: what a profiler lists)
        roof_launch_s, roof_updates, roof_note, full_eval = launch_s, updates_per_launch, None, None
        term_by_term = None
        if args.workload == "cfg3" and not args.single_region:      # (profiling runs keep to ONE kind of launch)
            # beside the default (the exact fast-forward of the two-valued sum: pass-free, priced against vector issue below): the TERM-BY-TERM pass
            # (exact_division = 1: one fp64 add per observation)
            t = A.Sampler(spec, chains=chains, seed=SEED, chain_offset=offset, device=dev_index, lanes_per_chain=1, steps_per_launch=20, exact_division=1)
            t.burn(40)
            t.burn(20)
            tv = chains * 20 * P / (t.launch_info()["kernel_ms"] * 1e-3)
            term_by_term = {"value": tv, "frac": tv * n_obs * ops_per_obs / FP64_VALU_PEAK, "kernel": t.launch_info()["kernel"],
                            "note": "options.exact_division = 1: the scalar jump-table pass, one fp64 add per observation (bit-identical to the fast-forward)"}
            t.close()
        if args.workload == "cfg4" and not args.group_local and not args.full_evaluation and not args.single_region:      # (profiling runs -- --single-region -- keep to ONE kernel)
            t = A.Sampler(spec, chains=chains, seed=SEED, chain_offset=offset, device=dev_index, lanes_per_chain=args.lanes, block_threads=args.block, steps_per_launch=args.steps_per_launch, full_evaluation=1)
            t.burn(2 * args.steps_per_launch)
            t.burn(3 * args.steps_per_launch)
            full_launch_s, full_updates = t.launch_info()["kernel_ms"] * 1e-3, chains * 3 * args.steps_per_launch * P
            full_eval = {"value": full_updates / full_launch_s, "frac": full_updates / full_launch_s * n_obs * ops_per_obs / FP64_VALU_PEAK, "kernel": t.launch_info()["kernel"]}
            if not str(li.get("kernel", "")).startswith("amwg_sweep_kernel"):
                roof_launch_s, roof_updates, kernel = full_launch_s, full_updates, "%s with options.full_evaluation = 1" % full_eval["kernel"]
                roof_note = "roofline = every evaluation passes over all the data (full_evaluation = 1), %.3g param-updates/s" % full_eval["value"]
            t.close()
        ckind = certified_kind(spec["model"], li["lanes_per_chain"], args.full_evaluation)
        certified = ckind is not None
        ops_note = OPS_NOTE[spec["model"]]
        if certified:
            # `value` and `frac` describe the same kernel: the certified pass and ITS operations per observation; the expression kernel is measured beside it
            if not args.single_region:
                few = 2 * args.steps_per_launch if ckind == "normal" else max(4, min(args.steps_per_launch, 10))
                t = A.Sampler(spec, chains=chains, seed=SEED, chain_offset=offset, device=dev_index, lanes_per_chain=li["lanes_per_chain"], block_threads=args.block, steps_per_launch=args.steps_per_launch, full_evaluation=1)
                t.burn(few)
                t.burn(few)
                fl = t.launch_info()
                fv = chains * few * P / (fl["kernel_ms"] * 1e-3)
                full_eval = {"value": fv, "frac": fv * n_obs * ops_per_obs / FP64_VALU_PEAK, "kernel": fl["kernel"], "lane_ops_per_obs": ops_per_obs,
                             "note": "options.full_evaluation = 1: the reference's expression, term by term, in every update (rounds 1-4's kernel)"}
                t.close()
            ops_per_obs, ops_note = (CERTIFIED_OPS_PER_OBS, CERTIFIED_NOTE) if ckind == "normal" else (CERTIFIED_GLM_OPS_PER_OBS, CERTIFIED_GLM_NOTE)
        traffic, traffic_src, traffic_alg, traffic_why_not = measured_traffic(chains, args.steps_per_launch, args.workload, li["lanes_per_chain"], args.group_local, kernel_id_of(version), kernel)
        lane_ops = roof_updates * n_obs * ops_per_obs / roof_launch_s
        if str(kernel).startswith("amwg_sweep_kernel"):
            # `value` and `frac` describe the same kernel: the sweep kernel against the arithmetic of its three passes per step
            lane_ops = sweep_lane_ops(roof_updates / roof_launch_s, P, n_obs)
            fe = full_eval or {"value": float("nan"), "frac": float("nan")}
            roof_note = SWEEP_NOTE % (kernel, P, n_obs, fe["value"], fe["frac"])
            ops_per_obs, ops_note = HIER_SWEEP_OPS, HIER_SWEEP_OPS_NOTE
        if args.group_local:
            lane_ops *= 2.0 / P          # two passes per step of P updates (see measure_other_config)
            label += " -- GROUP-LOCAL evaluation (opt-in; not the reference's operation schedule)"
        out = {
            "metric": "posterior draws/sec (= param-updates/sec)", "value": value, "unit": "param-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": (value / 8.0e4) if args.workload == "readme" else None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label,
                       "n_obs": n_obs, "chains_per_gpu": chains, "chains_total": total_chains, "components": P,
                       "draws_recorded_per_chain": rows, "thin": thin, "lanes_per_chain": li["lanes_per_chain"],
                       "block_threads": li["block_threads"], "grid_blocks": li["grid_blocks"], "lds_bytes": li["lds_bytes"],
                       "steps_per_launch": args.steps_per_launch, "launches_timed": launches,
                       "gather": ("amwg_comm_gather_draws: the library's RCCL communicator over the %d ranks (grouped ncclSend / ncclRecv to rank 0), inside the timed region" % world) if comm is not None
                                 else (("torch.distributed %s gather of recorded draws to rank 0" % ("rccl" if backend == "nccl" else backend)) if world > 1 else "none (1 GPU, --torch-gather or no communicator)"),
                       "rccl_ranks_seen": (comm_info or {}).get("rccl_ranks_seen"), "rccl_note": "ncclCommCount of the library's communicator (amwg_comm_info)",
                       "communicator_error": comm_error, "ranks": rank_devices},
            "timing": {"regions": len(regions), "reported": "median region", "region_ms": [r[0] * 1e3 for r in regions][:64],
                       "first_region_ms": regions[0][0] * 1e3, "min_region_ms": regions[order[0]][0] * 1e3, "max_region_ms": regions[order[-1]][0] * 1e3,
                       "steps_per_region": K, "note": "every region is exactly K steps between barrier + synchronize pairs; chains keep adapting across regions"},
            "roofline": {"bound": "fp64_valu", "achieved": lane_ops, "peak": FP64_VALU_PEAK, "unit": "fp64 lane-operations/s", "frac": lane_ops / FP64_VALU_PEAK,
                         "peak_note": "78.6 TFLOP/s fp64 vector (MI355X_MICROARCH.md) = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 3.93e13 lane-FMA/s",
                         "measured_peak": measured_peak, "frac_of_measured_peak": lane_ops / measured_peak,
                         "measured_peak_note": "amwg_fp64_peak: independent v_fma_f64 chains, no memory traffic, same process",
                         "lane_ops_per_obs": ops_per_obs, "lane_ops_note": ops_note,
                         "kernel": kernel + (" (certified decisions)" if certified else ""), "launch_ms": roof_launch_s * 1e3, "updates_per_launch": roof_updates,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_refused": traffic_why_not if traffic is None else None,
                         "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KB from separate rocprofv3 --pmc passes",
                         "algorithmic_bytes_per_launch": updates_per_launch * b_alg, "algorithmic_bytes_per_update": b_alg,
                         "traffic_ratio": (traffic / traffic_alg) if (traffic and traffic_alg) else None,
                         "traffic_note": "traffic and traffic_ratio are per launch of the PROFILED command (%d steps per launch, profiles/): measured HBM bytes / algorithmic bytes of that launch" % args.steps_per_launch,
                         "effective_hbm": {"achieved": eff_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": eff_gbps / HBM_PEAK_GBPS, "lds_resident": True,
                                           "note": "SURVEY.md section 8(d) contract figure: algorithmic bytes (one pass over the data per update) / launch time.  "
                                                   "The data vector is staged once per launch into LDS (L2/MALL for cfg5) and re-read from there, so this is an "
                                                   "EFFECTIVE rate that exceeds the HBM peak by design; it is not the roof this kernel runs against"},
                         "note": roof_note},
            "kernel_only_value": chains * K * P / (kernel_ms * 1e-3), "full_evaluation": full_eval,
            "posterior": {"mean": mean.tolist()[:8], "sd": sd.tolist()[:8], "data_mean": float(np.mean(x)), "data_sd": float(np.std(x, ddof=1)),
                          "note": "moments over the recorded draws of the last region on ALL ranks (all-reduce of per-rank sums for N > 1; after %d warm-up + %d timed steps)" % (W, K * (len(regions) - 1))},
        }
        if args.workload == "cfg3":
            # `value` and `frac` describe the SAME kernel (round-5 review, item 5): a pass-free update is bound by vector ISSUE -- the stepper's instructions per update
            # (committed rocprofv3 PMC pass of these kernel sources) x the measured update rate, against 1024 SIMDs x 2.4 GHz / 4 wave64 instructions per second
            vpu, vfile, why = profiled_valu_per_update("cfg3", kernel, kernel_id_of(version))
            rate = updates_per_launch / launch_s
            out["roofline"].update({"bound": "valu_issue", "achieved": (rate / 64.0) * vpu if vpu else None, "peak": VALU_ISSUE_PEAK, "unit": "wave64 vector instructions/s",
                                    "frac": ((rate / 64.0) * vpu / VALU_ISSUE_PEAK) if vpu else None, "frac_of_measured_peak": None, "valu_per_64_updates": vpu,
                                    "valu_profile": vfile, "valu_profile_refused": why, "lane_ops_per_obs": None,
                                    "lane_ops_note": "pass-free update: the exact fast-forward of the two-valued sum (csrc/amwg_twoval.h, ~log2 N binade steps) -- Philox, Leva's rnorm, two "
                                                     "logarithms and the bisection are all there is; the roof is vector issue, priced with the PMC pass of the same kernel sources"})
            out["term_by_term"] = term_by_term
        out["roofline"].update(sweep_roofline_units(lane_ops) if str(kernel).startswith("amwg_sweep_kernel") else
                               roofline_units(spec["model"], lane_ops, ops_per_obs) if not certified else
                               {"frac_issue": lane_ops / FP64_VALU_PEAK, "frac_survey_flops": (roof_updates / roof_launch_s) * n_obs * (3 if ckind == "normal" else 50) / FP64_FLOPS_PEAK,
                                "survey_flops_note": "the certified pass in flops (an FMA = 2) against 78.6 TFLOP/s: a sub and an fma per observation = 3 (Normal); 8 + 21 fmas = 50 (Poisson)"})
        import build_id
        out["library"] = {"version": version, "built_from_this_tree": ("build " + build_id.build_id()) in version and ("kernels " + build_id.kernel_id()) in version,
                          "note": "amwg_version(): a hash over every source of libamwg.so and one over the device sources + compiler flags (tools/build_id.py); "
                                  "built_from_this_tree compares them with the sources beside this bench.py"}
        if parity is not None:
            out["parity"] = parity
            fr = flip_rate_record(kernel_id_of(version))
            if fr is not None:
                out["parity"]["flip_rate"] = fr
        if world == 1 and args.workload == "cfg2" and not args.no_other_configs:
            out["other_configs"] = {}
            for name in ("cfg3", "cfg4", "cfg5", "cfg4_group_local"):
                try:
                    out["other_configs"][name] = measure_other_config(A, name.split("_")[0], dev_index, group_local=int(name.endswith("group_local")))
                except Exception as e:      # a failure here must not cost the headline line
                    out["other_configs"][name] = {"error": repr(e)}
            try:      # the opt-in third tier of the Normal family (amwg_options::sufficient_statistics), beside the headline's pass kernel and full_evaluation
                out["sufficient_statistics"] = measure_sufficient_statistics(A, spec, chains, dev_index)
            except Exception as e:
                out["sufficient_statistics"] = {"error": repr(e)}
            if not args.no_js:
                out["end_to_end_js"] = end_to_end_js(chains, n_obs)
            o = out["other_configs"]
            if "value" in o.get("cfg4", {}) and "value" in o.get("cfg4_group_local", {}):
                o["cfg4_group_local"]["speedup_over_cfg4"] = o["cfg4_group_local"]["value"] / o["cfg4"]["value"]
        if world == 1 and not args.no_cpu_baseline:
            ref = cpu_baseline_reference(args.workload)
            out["cpu_baseline"] = ref if ref is not None else cpu_baseline_port(spec)
            if ref is not None:
                out["cpu_baseline_port"] = cpu_baseline_port(spec, budget_s=3.0)
            port_rate = (out.get("cpu_baseline_port") or out["cpu_baseline"])["value"]
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(spec, port_rate)
            out["chains_equiv"] = value / out["cpu_baseline"]["value"]
        emit(out)
    s.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
