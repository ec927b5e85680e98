"""Shared helpers for the -m gpu parity tests (HIP path vs oracle / golden)."""
import numpy as np

import amwg_ctypes
import oracle_lib


def run_schedule(s, schedule):
    """Runs a golden-style schedule on a Sampler or OracleChain; returns list of draw arrays."""
    segs, thin = [], 1
    for seg in schedule:
        if seg["op"] == "burn":
            s.burn(seg["n"])
        elif seg["op"] == "stop":
            s.set_adapting(False)
        elif seg["op"] == "start":
            s.set_adapting(True)
        elif seg["op"] == "sample":
            thin = seg.get("thin", thin)
            segs.append(s.sample(seg["n"], thin))
    return segs


def run_schedule_many(samplers, schedule):
    """The same schedule on several samplers AT ONCE (each has its own stream: the launches of one-chain samplers overlap on the device)
    -> one list of draw arrays per sampler."""
    segs, thin = [[] for _ in samplers], 1
    for seg in schedule:
        if seg["op"] == "burn":
            for s in samplers:
                s.burn_async(seg["n"])
            for s in samplers:
                s.sync()
        elif seg["op"] == "stop":
            for s in samplers:
                s.set_adapting(False)
        elif seg["op"] == "start":
            for s in samplers:
                s.set_adapting(True)
        elif seg["op"] == "sample":
            thin = seg.get("thin", thin)
            for s in samplers:
                s.sample_async(seg["n"], thin)
            for k, s in enumerate(samplers):
                segs[k].append(s.fetch_draws())
    return segs


def run_schedules_concurrently(jobs, max_threads=48):
    """jobs: [(sampler, schedule)] with DIFFERENT schedules -> [list of draw arrays per job].  One host thread per sampler (ctypes releases the GIL; every
    sampler has its own stream): a one-lane chain at N = 5e4 is a single wavefront for a minute, and a dozen of those fit the chip side by side."""
    import concurrent.futures
    if not jobs:
        return []
    with concurrent.futures.ThreadPoolExecutor(min(max_threads, len(jobs))) as ex:
        return list(ex.map(lambda j: run_schedule(j[0], j[1]), jobs))


def assert_chain_equals_oracle(gpu, local, orc, gpu_segs, orc_segs):
    """Bit-exact comparison of local chain `local` of a GPU sampler with an oracle chain run in the same order."""
    for g, o in zip(gpu_segs, orc_segs):
        assert g[:, :, local].tobytes() == np.ascontiguousarray(o).tobytes()
    gi, oi = gpu.info(), orc.info()
    for k in ("accepts", "inbounds", "batch_count", "acceptance_count", "iterations_since_adaption"):
        assert gi[k][:, local].tolist() == oi[k].tolist(), k
    assert gi["prop_log_scale"][:, local].tobytes() == oi["prop_log_scale"].tobytes()
    assert gpu.state()[:, local].tobytes() == orc.state().tobytes()
    d = gpu.diag()
    assert int(d["uniforms"][local]) == orc.uniforms()
    assert d["named_order"][local].tolist() == orc.named_order().tolist()
    assert np.float64(d["log_post"][local]).tobytes() == np.float64(orc.log_post()).tobytes()
