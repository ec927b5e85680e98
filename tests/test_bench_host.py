"""bench.py's host logic that needs no GPU: which committed profile a roofline figure may take its HBM traffic from."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_kernel_names_are_compared_in_one_spelling():
    import bench
    assert bench.kernel_base_name("void amwg::amwg_sweep_kernel<amwg::HierNormalModel, 512>(amwg::StepArgs)") == "amwg_sweep_kernel<HierNormalModel,512>"
    assert bench.kernel_base_name("amwg_step_kernel<HierNormalModel,64,512> with options.full_evaluation = 1") == "amwg_step_kernel<HierNormalModel,64,512>"
    assert bench.kernel_base_name("void amwg::amwg_gl_kernel<amwg::HierGlModel, 512>(amwg::StepArgs)") == "amwg_gl_kernel<HierGlModel,512>"
    assert bench.kernel_base_name("amwg_user_step") == "amwg_user_step"


def test_traffic_comes_from_a_profile_of_the_same_kernel_and_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic: the newest committed profile of the same workload, chains, steps per launch AND kernel -- cfg4's roofline figure is the
    full-evaluation step kernel, not the sweep kernel that produces `value` -- and only if its kernel id is the library's (a profile of other
    kernel sources is refused, with the reason)."""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()

    def put(name, kernel, kid, cmd="bench.py --workload cfg4 --weak", traffic=1.0e7):
        (prof / name).write_text(json.dumps({"workload": "cfg4", "command": cmd, "kernel": kernel, "kernel_id": kid, "chains": 2048, "steps_per_launch": 100,
                                             "hbm_traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": 6.3e11}))
    put("r09_cfg4_summary.json", "void amwg::amwg_sweep_kernel<amwg::HierNormalModel, 512>(amwg::StepArgs)", "abc", traffic=4.1e7)
    put("r09_cfg4full_summary.json", "void amwg::amwg_step_kernel<amwg::HierNormalModel, 64, 512>(amwg::StepArgs)", "abc", cmd="bench.py --workload cfg4 --weak --full-evaluation", traffic=3.7e7)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, src, alg, why = bench.measured_traffic(2048, 100, "cfg4", 64, False, "abc", "amwg_step_kernel<HierNormalModel,64,512> with options.full_evaluation = 1")
    assert t == 3.7e7 and src.endswith("r09_cfg4full_summary.json") and why is None
    t, src, alg, why = bench.measured_traffic(2048, 100, "cfg4", 64, False, "abc", "amwg_sweep_kernel<HierNormalModel,512>")
    assert t == 4.1e7 and src.endswith("r09_cfg4_summary.json")
    t, src, alg, why = bench.measured_traffic(2048, 100, "cfg4", 64, False, "other", "amwg_sweep_kernel<HierNormalModel,512>")
    assert t is None and "refused" in why and "abc" in why and "other" in why
    t, src, alg, why = bench.measured_traffic(4096, 100, "cfg4", 64, False, "abc", "amwg_sweep_kernel<HierNormalModel,512>")
    assert t is None and "no profile" in why


def _fat_record():
    """a record at least as large as the one the round-4 driver could not parse (profiles/r04_bench_default.json, 19 KB on one line), made larger"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    d["timing"]["region_ms"] = [1.2345678901234] * 400
    d["parity"]["flip_rate"]["per_run"] = d["parity"]["flip_rate"]["per_run"] * 8
    d["roofline"]["note"] = "x" * 5000
    return d


def test_the_stdout_line_stays_under_4_kb_whatever_the_record_holds(tmp_path):
    """Round-4 review: BENCH_r04.json had parsed = null because the line had grown to 19 KB.  stdout carries compact_line(record) <= 4 KB -- the
    contract's fields, the roofline and cpu_baseline objects, three parity booleans, value + frac of the other configs -- and the rest goes to the side file."""
    import bench
    d = _fat_record()
    assert len(json.dumps(d)) > 19000
    text = bench.compact_line(d, str(tmp_path / "bench_detail.json"))
    assert "\n" not in text and len(text) <= bench.LINE_LIMIT == 4096 and len(text) < 8192
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line and (line[k] == d[k] or abs(line[k] - d[k]) <= 1e-6 * abs(d[k])), k
    assert line["config"]["workload"] == d["config"]["workload"] and line["config"]["chains_per_gpu"] == 65536 and "rccl_ranks_seen" in line["config"]
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")) <= set(r) and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["effective_hbm"]["frac"] > 1 and r["effective_hbm"]["lds_resident"] is True
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "param-updates/s" and 0 < len(c["sample"]) <= 160
    assert line["parity"]["accept_counts_identical"] and line["parity"]["draws_bit_identical"] and line["parity"]["final_state_bit_identical"]
    assert set(line["other_configs"]) == {"cfg3", "cfg4", "cfg5", "cfg4_group_local"}
    assert all(o["value"] > 0 and 0 < o["frac"] < 1 and o["parity_ok"] for o in line["other_configs"].values())
    assert "dropped_for_size" not in line


def test_a_line_that_would_still_be_too_long_sheds_optional_parts_not_the_contract(tmp_path):
    import bench
    d = _fat_record()
    d["other_configs"] = {"cfg%d" % i: dict(d["other_configs"]["cfg4"]) for i in range(60)}
    text = bench.compact_line(d, None)
    line = json.loads(text)
    assert len(text) <= 4096 and "other_configs" in line["dropped_for_size"] and line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_emit_writes_one_line_to_stdout_the_record_to_the_side_file_and_nothing_json_like_to_stderr(tmp_path):
    import subprocess
    side = tmp_path / "bench_detail.json"
    code = ("import json, sys; sys.path.insert(0, %r); import bench; d = json.load(open(%r)); bench.claim_stdout(); print('a library banner'); bench.emit(d)"
            % (ROOT, os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, AMWG_BENCH_DETAIL=str(side)), timeout=120)
    assert p.returncode == 0, p.stderr[-1000:]
    assert p.stdout.count("\n") == 1 and len(p.stdout) <= 4097 and json.loads(p.stdout)["value"] > 0
    assert "a library banner" in p.stderr and not any(ln.startswith("{") for ln in p.stderr.splitlines())
    assert json.load(open(side))["timing"]["regions"] >= 3


def test_gpus_n_without_a_launcher_prints_the_not_measured_line_on_a_box_without_n_devices(tmp_path):
    """Round-4 review: `python3 bench.py --gpus 8` outside torch.distributed.run was a SystemExit.  Here (no GPU): rc 0, one line, value null."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AMWG_BENCH_DETAIL"] = str(tmp_path / "d.json")
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-1000:]
    line = json.loads(p.stdout)
    assert p.stdout.count("\n") == 1 and line["value"] is None and line["n_gpus"] == 2 and line["steps"] == 20 and "not measured" in line["note"]


def run_bench_under_the_launcher(tmp_path, n=2, port=29631, extra_env=None, extra_args=()):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 ... bench.py --gpus n` -- the command the driver runs for its scaling
    curve -- on a box with fewer than n visible devices: every rank leaves with rc 0 and rank 0 prints ONE line, value null, "not measured".  -> the line"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AMWG_BENCH_DETAIL"] = str(tmp_path / "d.json")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2"] + list(extra_args), capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-1500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    return json.loads(lines[0])


def test_two_ranks_under_torch_distributed_run_on_a_box_without_two_devices_say_not_measured(tmp_path):
    """Round-5 review, item 7: the launcher-started two-process path must still hold where two devices are not visible (here: none)."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    line = run_bench_under_the_launcher(tmp_path)
    assert line["value"] is None and line["n_gpus"] == 2 and line["steps"] == 5 and "not measured" in line["note"]


def test_flip_rate_of_a_campaign_run_with_other_kernels_is_refused(tmp_path, monkeypatch):
    """parity.flip_rate quotes the committed campaign only when it ran the kernels that are being timed (round-4 review: two kept bench lines quoted
    a campaign of earlier kernels); the newest campaign by ROUND NAME is the one looked at."""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    rec = {"version": "amwg-mi355x 0.4 (gfx950) build aaaaaaaaaaaa kernels 111111111111", "runs": [], "decisions_total": 10 ** 10, "first_flips_total": 1,
           "flips_per_1e9": 0.1, "upper_95_per_1e9": 0.5,
           "reference_order": {"decisions_total": 2 * 10 ** 10, "chains_differing": 0, "log_post_differs": False, "geometries": [{"lanes_per_chain": 64, "summation_order": 1}] * 3}}
    (prof / "r04_flip_rate.json").write_text(json.dumps(dict(rec, version=rec["version"].replace("111111111111", "000000000000"))))
    (prof / "r05_flip_rate.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    ok = bench.flip_rate_record("111111111111")
    assert ok["source"].endswith("r05_flip_rate.json") and ok["flips_per_1e9"] == 0.1 and "refused" not in ok
    # (round 5: the kernels that decide in the reference's order are reported apart -- chains differing of how many decisions -- and reach the stdout line without their geometry list)
    assert ok["reference_order"]["chains_differing"] == 0 and len(ok["reference_order"]["geometries"]) == 3
    line = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "parity": {"accept_counts_identical": True, "flip_rate": ok}}))
    assert line["parity"]["flip_rate"]["reference_order"] == {"decisions_total": 2 * 10 ** 10, "chains_differing": 0, "log_post_differs": False}
    no = bench.flip_rate_record("222222222222")
    assert "refused" in no and "111111111111" in no["refused"] and "222222222222" in no["refused"] and "flips_per_1e9" not in no
    line = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "parity": {"accept_counts_identical": True, "flip_rate": no}}))
    assert "refused" in line["parity"]["flip_rate"]


def test_sweep_kernel_roofline_counts_one_certified_pass_per_step():
    import bench
    # 3.4e9 updates/s of the 34-component model over 1e4 observations: 1e8 steps/s x 1 pass (the sweep's sums of squares; mu and sigma read no data) x 1e4 x 2 operations
    ops = bench.sweep_lane_ops(3.4e9, 34, 10_000)
    assert abs(ops - 3.4e9 / 34 * 1 * 10_000 * 2) < 1 and 0.04 < ops / bench.FP64_VALU_PEAK < 0.08
