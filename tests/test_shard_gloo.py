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
    glist = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    shard.gather_draws(dist, mine, glist, rank)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the bench's max-over-ranks timing reduction
    if rank == 0:
        q.put((shard.merge_gathered(glist).numpy(), float(t[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    world, total, rows = 2, 6, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    assert merged.tobytes() == _draws_for(0, total, rows).tobytes()


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
