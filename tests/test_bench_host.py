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
