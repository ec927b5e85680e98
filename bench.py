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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

N_OBS = 10_000
CHAINS_PER_GPU = 65_536
SEED = 20260925
DATA_SEED = 20260925
B_ALG_PER_UPDATE = N_OBS * 8 + 8 * 2 + 8     # 80 024 B: data once + state read + draw write
HBM_PEAK_GBPS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK = 78.6e12 / 2                 # lane-FMA/s: 256 CU * 4 SIMD * 16 lanes/clk * 2.4 GHz


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
    "cfg5": ("pois_glm", 50_000, 8_192, 50_000 * (7 * 8 + 8 + 8) + 8 * 9 + 8, 127, "BASELINE.json configs[4]: Poisson GLM + int change point, 5e4 obs, 8192 chains per GPU (65536 over 8)"),
}


# what the fp64 lane-operations per observation of each family are (the roofline's unit of arithmetic)
OPS_NOTE = {
    "normal": "8 = sub, mul, 4-operation correctly rounded quotient (amwg_div.h: mul, fma, fma, fma), sub, add; IEEE '/' would be 17",
    "hier_normal": "8 = sub, mul, 4-operation correctly rounded quotient, sub, add (the gather of theta[g_i] is an LDS read, not arithmetic)",
    "beta_bern": "1 = the fp64 add of the term-by-term pass (the observation selects WHICH register is added, on the scalar unit)",
    "pois_glm": "127 = VALU instructions ISSUED per observation (rocprofv3, profiles/r01g): 7 fma of the linear predictor, V8's exp and log "
                "(fdlibm, ~50 each incl. their correctly rounded quotients), 4 for the density; an issue count, not a minimal operation count",
}


def other_spec(name, exp):
    import model_spec
    fam, n_obs = OTHER_WORKLOADS[name][0], OTHER_WORKLOADS[name][1]
    return model_spec.build_spec(fam, model_spec.make_data(fam, n_obs, DATA_SEED, G=32, exp=exp))


def measured_traffic(chains, steps_per_launch, workload="cfg2", lanes=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_summary.json, written by
    tools/profile.sh + tools/summarize_profile.py for this same command); None if no matching profile."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")), reverse=True):
        try:
            p = json.load(open(f))
        except (OSError, ValueError):
            continue
        if p.get("workload", "cfg2") != workload:
            continue
        if lanes is not None and (", %d>" % lanes) not in p.get("kernel", ""):      # the profile must be of the same kernel instantiation
            continue
        if p.get("chains") == chains and p.get("steps_per_launch") == steps_per_launch and p.get("hbm_traffic_bytes_per_launch"):
            return p["hbm_traffic_bytes_per_launch"], os.path.relpath(f, ROOT), p.get("algorithmic_bytes_per_launch")
    return None, None, None


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


def parity_gate(A, spec, timed_lanes):
    """next to the speed, the proof that this build reproduces the reference.  The first and the
    last chain id of the bench job, same seed and data, against the seeded run of the UNMODIFIED reference stored in
    tests/golden/cfg2_full.json (a committed fixture; nothing here reads /root/reference): with one lane per chain (the
    reference's summation order) every draw must be bit-identical; with the lane count of the TIMED configuration, if that
    differs, the accept counts must still be identical.  If the timed configuration is not the one-lane one, the whole job is
    also timed in reference order for 100 steps."""
    import golden_io
    gold = golden_io.load("cfg2_full")

    def run(lanes):
        ok_acc = ok_state = ok_draws = True
        for rec in gold["chains"]:
            s = A.Sampler(spec, chains=1, seed=gold["case"]["seed"], chain_offset=rec["chain"], lanes_per_chain=lanes)
            draws = None
            for seg in gold["case"]["schedule"]:
                if seg["op"] == "burn":
                    s.burn(seg["n"])
                else:
                    draws = s.sample(seg["n"], seg.get("thin", 1))
            want = np.array(rec["samples"][0]["draws"], dtype=np.float64)
            ok_draws = ok_draws and np.ascontiguousarray(draws[: want.shape[0], :, 0]).tobytes() == want.tobytes()
            ok_acc = ok_acc and s.info()["accepts"][:, 0].tolist() == rec["accepts"]
            ok_state = ok_state and s.state()[:, 0].tolist() == rec["final_state"]
            s.close()
        return bool(ok_draws), bool(ok_acc), bool(ok_state)

    ok_draws, ok_acc, ok_state = run(1)
    out = {"golden": "tests/golden/cfg2_full.json (seeded run of the unmodified reference, chains 0 and 65535, burn 500 + sample 500)",
           "lanes_per_chain": 1, "draws_bit_identical": ok_draws, "accept_counts_identical": ok_acc, "final_state_bit_identical": ok_state,
           "timed_lanes_per_chain": timed_lanes, "timed_configuration_is_reference_order": timed_lanes == 1}
    if timed_lanes != 1:
        d2, a2, _ = run(timed_lanes)
        out["timed_lanes_accept_counts_identical"] = a2
        out["timed_lanes_draws_bit_identical"] = d2
        s = A.Sampler(spec, chains=CHAINS_PER_GPU, seed=SEED, lanes_per_chain=1, steps_per_launch=100)
        s.burn(300)
        s.burn(100)
        out["reference_order_value"] = CHAINS_PER_GPU * 100 * spec["P"] / (s.launch_info()["kernel_ms"] * 1e-3)
        s.close()
        out["note"] = ("reference_order_value = param-updates/s of the same 65536-chain job with one lane per chain, i.e. every chain in the "
                       "reference's exact summation order (bit-identical draws); the timed configuration splits a chain's sum over %d lanes" % timed_lanes)
    else:
        out["note"] = "the timed configuration runs one lane per chain: every draw of every chain is the reference's, bit for bit"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--thin", type=int, default=10, help="record every thin-th draw inside the timed region")
    ap.add_argument("--chains-per-gpu", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--steps-per-launch", type=int, default=100,
                    help="steps fused into one kernel launch; warm-up and timed steps use the same launch size so the "
                         "per-launch time bench.py reports is comparable with rocprofv3's per-kernel average")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the timed K-step region until this much time is on the clock (median reported)")
    ap.add_argument("--single-region", action="store_true", help="time the K-step region once (profiling runs)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "readme"],
                    help="cfg2 (default) is the bench line; the others measure the remaining BASELINE.json configs")
    args = ap.parse_args()

    import torch
    import amwg_ctypes as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Development switches (not used by the driver): AMWG_BENCH_BACKEND=gloo + AMWG_BENCH_ONE_DEVICE=1 run the
    # N-rank code path on a box with a single GPU (ranks share cuda:0, gather goes through host memory).
    backend = os.environ.get("AMWG_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("AMWG_BENCH_ONE_DEVICE") == "1" else local_rank
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
    offset, _ = chain_shard(rank, world, chains * world)
    s = A.Sampler(spec, chains=chains, seed=SEED, chain_offset=offset, device=dev_index,
                  lanes_per_chain=args.lanes, block_threads=args.block, steps_per_launch=args.steps_per_launch)
    P, K, W, thin = spec["P"], args.steps, args.warmup, max(1, args.thin)
    rows = -(-K // thin)
    draws = torch.empty((rows, P, chains), dtype=torch.float64, device="cuda")
    gathered = [torch.empty(draws.shape, dtype=draws.dtype, device=coll_dev) for _ in range(world)] if (world > 1 and rank == 0) else None

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
        if dist is not None:
            gather_draws(dist, draws if coll_dev == "cuda" else draws.cpu(), gathered, rank, equal_sizes=True)
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

    if rank == 0:
        total_chains = chains * world
        value = total_chains * K * P / dt
        launches = max(1, li["n_launches"])
        launch_s = kernel_ms * 1e-3 / launches
        updates_per_launch = chains * (K / launches) * P
        eff_gbps = updates_per_launch * b_alg / launch_s / 1e9
        mean, sd = (s.moments() if dist is None else (pooled[0].cpu().numpy(), pooled[1].cpu().numpy()))
        pm = pooled[0].cpu().numpy()
        assert dist is not None or np.allclose(pm, mean, rtol=1e-10, atol=0), "library moments differ from the pooled restatement"
        measured_peak = A.fp64_peak(dev_index)      # register-only fma kernel: what the chip sustains under fp64 load
        x = spec["data"]["x"]
        traffic, traffic_src, traffic_alg = measured_traffic(chains, args.steps_per_launch, args.workload, li["lanes_per_chain"])
        kname = {"normal": "NormalModel", "beta_bern": "BetaBernModel", "hier_normal": "HierNormalModel", "pois_glm": "PoisGlmModel"}[spec["model"]]
        kernel = "amwg_step_kernel<%s,%d>" % (kname, li["lanes_per_chain"])
        roof_launch_s, roof_updates, roof_note = launch_s, updates_per_launch, None
        if args.workload == "cfg3":
            # the headline value uses the exact fast-forward of the two-valued sum, which does not stream the data at all; the
            # roofline figure is the TERM-BY-TERM pass (exact_division = 1: one fp64 add per observation), measured on the side
            t = A.Sampler(spec, chains=chains, seed=SEED, chain_offset=offset, device=dev_index, lanes_per_chain=1, steps_per_launch=20, exact_division=1)
            t.burn(40)
            t.burn(20)
            roof_launch_s, roof_updates = t.launch_info()["kernel_ms"] * 1e-3, chains * 20 * P
            kernel = "amwg_step_kernel<BetaBernModel,1> term-by-term pass (exact_division = 1)"
            roof_note = ("roofline = the term-by-term pass (scalar jump-table kernel, 1 fp64 add per observation), %.3g param-updates/s; `value` is the exact "
                         "fast-forward of the same sum (bit-identical, ~log2(N) binade steps instead of N additions), which has no meaningful roofline"
                         % (roof_updates / roof_launch_s))
            t.close()
        lane_ops = roof_updates * n_obs * ops_per_obs / roof_launch_s
        out = {
            "metric": "posterior draws/sec (= param-updates/sec)", "value": value, "unit": "param-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": (value / 8.0e4) if args.workload == "readme" else None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label,
                       "n_obs": n_obs, "chains_per_gpu": chains, "chains_total": total_chains, "components": P,
                       "draws_recorded_per_chain": rows, "thin": thin, "lanes_per_chain": li["lanes_per_chain"],
                       "block_threads": li["block_threads"], "grid_blocks": li["grid_blocks"], "lds_bytes": li["lds_bytes"],
                       "steps_per_launch": args.steps_per_launch, "launches_timed": launches,
                       "gather": ("%s gather of recorded draws to rank 0" % ("rccl" if backend == "nccl" else backend)) if world > 1 else "none (1 GPU)"},
            "timing": {"regions": len(regions), "reported": "median region", "region_ms": [r[0] * 1e3 for r in regions][:64],
                       "first_region_ms": regions[0][0] * 1e3, "min_region_ms": regions[order[0]][0] * 1e3, "max_region_ms": regions[order[-1]][0] * 1e3,
                       "steps_per_region": K, "note": "every region is exactly K steps between barrier + synchronize pairs; chains keep adapting across regions"},
            "roofline": {"bound": "fp64_valu", "achieved": lane_ops, "peak": FP64_VALU_PEAK, "unit": "fp64 lane-operations/s", "frac": lane_ops / FP64_VALU_PEAK,
                         "peak_note": "78.6 TFLOP/s fp64 vector (MI355X_MICROARCH.md) = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 3.93e13 lane-FMA/s",
                         "measured_peak": measured_peak, "frac_of_measured_peak": lane_ops / measured_peak,
                         "measured_peak_note": "amwg_fp64_peak: independent v_fma_f64 chains, no memory traffic, same process",
                         "lane_ops_per_obs": ops_per_obs, "lane_ops_note": OPS_NOTE[spec["model"]],
                         "kernel": kernel, "launch_ms": roof_launch_s * 1e3, "updates_per_launch": roof_updates,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_unit": "HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE) KB from separate rocprofv3 --pmc passes",
                         "algorithmic_bytes_per_launch": updates_per_launch * b_alg, "algorithmic_bytes_per_update": b_alg,
                         "traffic_ratio": (traffic / traffic_alg) if (traffic and traffic_alg) else None,
                         "traffic_note": "traffic and traffic_ratio are per launch of the PROFILED command (%d steps per launch, profiles/): measured HBM bytes / algorithmic bytes of that launch" % args.steps_per_launch,
                         "effective_hbm": {"achieved": eff_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": eff_gbps / HBM_PEAK_GBPS, "lds_resident": True,
                                           "note": "SURVEY.md section 8(d) contract figure: algorithmic bytes (one pass over the data per update) / launch time.  "
                                                   "The data vector is staged once per launch into LDS (L2/MALL for cfg5) and re-read from there, so this is an "
                                                   "EFFECTIVE rate that exceeds the HBM peak by design; it is not the roof this kernel runs against"},
                         "note": roof_note},
            "kernel_only_value": chains * K * P / (kernel_ms * 1e-3),
            "posterior": {"mean": mean.tolist()[:8], "sd": sd.tolist()[:8], "data_mean": float(np.mean(x)), "data_sd": float(np.std(x, ddof=1)),
                          "note": "moments over the recorded draws of the last region on ALL ranks (all-reduce of per-rank sums for N > 1; after %d warm-up + %d timed steps)" % (W, K * (len(regions) - 1))},
        }
        if world == 1 and args.workload == "cfg2":
            out["parity"] = parity_gate(A, spec, li["lanes_per_chain"])
        if world == 1 and not args.no_cpu_baseline:
            ref = cpu_baseline_reference(args.workload)
            out["cpu_baseline"] = ref if ref is not None else cpu_baseline_port(spec)
            if ref is not None:
                out["cpu_baseline_port"] = cpu_baseline_port(spec, budget_s=3.0)
            port_rate = (out.get("cpu_baseline_port") or out["cpu_baseline"])["value"]
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(spec, port_rate)
            out["chains_equiv"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
