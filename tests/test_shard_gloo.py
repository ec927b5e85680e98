"""world_size-2 gloo test (CPU) of the N>1 path: shard offsets, the gather, and the merge.

Each rank produces the draws of ITS chain shard with the CPU oracle (standing in for the GPU
kernel, which is checked against the same oracle under -m gpu) and the gathered+merged result
must equal a single-process run over all chains -- i.e. sharding changes nothing.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _draws_for(offset, count, rows):
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "bayes.js_amd")]
    import model_spec
    import oracle_lib
    spec = model_spec.build_spec("normal", model_spec.make_data("normal", 50, 7))
    out = np.empty((rows, 2, count))
    for c in range(count):
        ch = oracle_lib.OracleChain(spec, 123, offset + c)
        ch.burn(20)
        out[:, :, c] = ch.sample(rows, 1)
    return out


def _worker(rank, world, port, total, rows, q):
    sys.path[:0] = [os.path.join(ROOT, "bayes.js_amd")]
    import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = shard.chain_shard(rank, world, total)
    mine = torch.from_numpy(_draws_for(off, cnt, rows))
    if rank == 0:          # unequal shards: gather by point-to-point, as shard.gather_draws does when the sizes differ
        glist = [torch.empty((rows, mine.shape[1], shard.chain_shard(r, world, total)[1]), dtype=mine.dtype) for r in range(world)]
    else:
        glist = None
    shard.gather_draws(dist, mine, glist, rank)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the bench's max-over-ranks timing reduction
    mean, sd = shard.pooled_moments(dist, mine)       # the collective summaries: every rank ends up with the pooled statistics
    rhat, ess = shard.pooled_convergence(dist, mine)
    if rank == 0:
        q.put((shard.merge_gathered(glist).numpy(), float(t[0]), mean.numpy(), sd.numpy(), rhat.numpy(), ess.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    world, total, rows = 2, 7, 8                      # 7 chains over 2 ranks: unequal shards (4 + 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, tmax, mean, sd, rhat, ess = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    want = _draws_for(0, total, rows)
    assert merged.tobytes() == want.tobytes()
    # pooled summaries through the all-reduces == numpy on the pooled draws == the one-process formulas without a collective
    flat = np.moveaxis(want, 1, 0).reshape(want.shape[1], -1)
    np.testing.assert_allclose(mean, flat.mean(axis=1), rtol=1e-13)
    np.testing.assert_allclose(sd, flat.std(axis=1, ddof=1), rtol=1e-12)
    sys.path[:0] = [os.path.join(ROOT, "bayes.js_amd")]
    import shard
    r1, e1 = shard.pooled_convergence(None, torch.from_numpy(want))
    np.testing.assert_allclose(rhat, r1.numpy(), rtol=1e-12)
    np.testing.assert_allclose(ess, e1.numpy(), rtol=1e-11)
    half = rows // 2                                  # split-R-hat restated with numpy (BDA3 section 11.4)
    hm = np.stack([want[:half].mean(axis=0), want[half:2 * half].mean(axis=0)])          # [2][P][C]
    hv = np.stack([want[:half].var(axis=0, ddof=1), want[half:2 * half].var(axis=0, ddof=1)])
    W = hv.mean(axis=(0, 2))
    B_over_n = np.moveaxis(hm, 1, 0).reshape(hm.shape[1], -1).var(axis=1, ddof=1)
    np.testing.assert_allclose(rhat, np.sqrt(((half - 1) / half * W + B_over_n) / W), rtol=1e-12)


def test_chain_shard_covers_everything_once():
    sys.path[:0] = [os.path.join(ROOT, "bayes.js_amd")]
    import shard
    for total in (1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                off, cnt = shard.chain_shard(r, world, total)
                assert off == nxt
                nxt += cnt
            assert nxt == total
