"""-m gpu: summaries over the shards of one job (amwg_group_*: per-device reductions combined with an RCCL all-reduce; shards on
one device are summed there first) against the single sampler that runs all the chains, and against numpy on the pooled draws."""
import numpy as np
import pytest

import amwg_ctypes as A
import model_spec

pytestmark = pytest.mark.gpu


def _run(spec, chains, offset, seed=31):
    s = A.Sampler(spec, chains=chains, seed=seed, chain_offset=offset, lanes_per_chain=1)
    s.burn(150)
    return s, s.sample(120, 3)          # 40 kept draws


@pytest.mark.parametrize("family,n_obs", [("normal", 400), ("hier_small", 0)])
def test_group_summaries_of_unequal_shards_equal_the_single_sampler(family, n_obs):
    if family == "normal":
        spec = model_spec.build_spec("normal", model_spec.make_data("normal", n_obs, 5))
    else:
        spec = model_spec.build_spec("hier_normal", model_spec.make_data("hier_normal", 300, 5, G=6))
    C, cut1, cut2 = 200, 72, 136
    whole, dw = _run(spec, C, 0)
    parts = [_run(spec, cut1, 0), _run(spec, cut2 - cut1, cut1), _run(spec, C - cut2, cut2)]
    shards = [p[0] for p in parts]
    pooled = np.concatenate([p[1] for p in parts], axis=2)
    assert pooled.tobytes() == dw.tobytes()                      # sharding does not change a draw (global chain ids key the RNG)
    probs = [0.0, 0.025, 0.25, 0.5, 0.975, 1.0]
    m1, sd1 = whole.moments()
    m3, sd3 = A.group_moments(shards)
    np.testing.assert_allclose(m3, m1, rtol=1e-13)
    np.testing.assert_allclose(sd3, sd1, rtol=1e-11)
    flat = np.moveaxis(dw, 1, 0).reshape(dw.shape[1], -1)
    np.testing.assert_allclose(m3, flat.mean(axis=1), rtol=1e-12)
    np.testing.assert_allclose(sd3, flat.std(axis=1, ddof=1), rtol=1e-10)
    r1, e1 = whole.convergence()
    r3, e3 = A.group_convergence(shards)
    np.testing.assert_allclose(r3, r1, rtol=1e-10)
    np.testing.assert_allclose(e3, e1, rtol=1e-9)
    assert A.group_quantiles(shards, probs).tobytes() == whole.quantiles(probs).tobytes()    # the same multiset, sorted
    # a group of one still goes through the (one-rank) RCCL all-reduce
    m0, sd0 = A.group_moments([whole])
    np.testing.assert_allclose(m0, m1, rtol=1e-13)
    np.testing.assert_allclose(sd0, sd1, rtol=1e-11)
    r0, e0 = A.group_convergence([whole])
    np.testing.assert_allclose(r0, r1, rtol=1e-10)
    assert A.group_quantiles([whole], probs).tobytes() == whole.quantiles(probs).tobytes()
    for s in shards + [whole]:
        s.close()


def test_group_calls_reject_mismatched_shards():
    spec = model_spec.build_spec("normal", model_spec.make_data("normal", 100, 5))
    a, _ = _run(spec, 8, 0)
    b = A.Sampler(spec, chains=8, seed=31, chain_offset=8, lanes_per_chain=1)
    with pytest.raises(A.AmwgError, match="no sample"):
        A.group_moments([a, b])
    b.burn(10)
    b.sample(30, 3)                      # 10 kept draws, a has 40
    with pytest.raises(A.AmwgError, match="kept"):
        A.group_moments([a, b])
    a.close()
    b.close()


def test_gather_at_sample_collection_puts_every_shard_block_on_the_root_device():
    """amwg_group_gather_draws (north_star's "RCCL gather at sample collection", one process): three shards -- here on one device, so their
    blocks are copied; on distinct devices they travel by grouped ncclSend / ncclRecv -- stand back to back in shard order on the root's
    device and arrive on the host in one copy, equal to what each shard returns by itself."""
    data = model_spec.make_data("normal", 500, 3)
    spec = model_spec.build_spec("normal", data)
    counts, off, shards = [300, 300, 201], 0, []
    for c in counts:
        shards.append(A.Sampler(spec, chains=c, seed=9, chain_offset=off))
        off += c
    for s in shards:
        s.burn(40)
        s.sample_async(30, 3)
    for s in shards:
        s.sync()
    blocks, offsets = A.group_gather_draws(shards, root=1)
    assert offsets == [0, 10 * 2 * 300, 10 * 2 * 600]
    for s, b in zip(shards, blocks):
        assert b.tobytes() == s.fetch_draws().tobytes()
    info = A.group_comm_info(shards)
    assert info["rccl_ranks_seen"] == 1 and info["devices"] == [0]
    for s in shards:
        s.close()


def test_one_process_per_device_communicator_with_a_single_rank():
    """amwg_comm_*: the communicator a one-process-per-device host builds from a shared id (bench.py --gpus N under torch.distributed.run).  A
    one-GPU box can only run it with one rank: id -> ncclCommInitRank -> what RCCL reports about it -> gather (the rank's own block lands at
    offset 0 of the root's buffer) -> the all-reduced moments equal the sampler's own."""
    import torch
    data = model_spec.make_data("normal", 400, 4)
    spec = model_spec.build_spec("normal", data)
    s = A.Sampler(spec, chains=777, seed=5)
    s.burn(50)
    s.sample_async(40, 2)
    s.sync()
    comm = A.Comm(1, 0, 0, lambda ident: ident)
    assert comm.info() == {"rccl_ranks_seen": 1, "rank": 0, "device": 0}
    n = 20 * 2 * 777
    dst = torch.full((n + 8,), -1.0, dtype=torch.float64, device="cuda")
    assert comm.gather_draws(s, 0, dst.data_ptr(), n * 8) == [n]
    torch.cuda.synchronize()
    assert dst[:n].cpu().numpy().tobytes() == s.fetch_draws().tobytes() and float(dst[n]) == -1.0
    m, sd = comm.moments(s)
    m0, sd0 = s.moments()
    assert np.allclose(m, m0, rtol=1e-12) and np.allclose(sd, sd0, rtol=1e-10)
    with pytest.raises(A.AmwgError, match="holds"):
        comm.gather_draws(s, 0, dst.data_ptr(), 16)
    comm.close()
    s.close()
