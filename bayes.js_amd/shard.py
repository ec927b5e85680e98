"""Chain sharding across GPUs -- host logic shared by bench.py and the tests.

Chains are independent (the reference has one chain; many chains = many independent samplers),
so the path shards with NO data-path collective: rank r owns the contiguous global chain ids
[offset, offset+count); the read-only data vector is replicated; Philox is keyed by the GLOBAL
chain id, so draws do not depend on the number of GPUs.  The only exchange is the gather of the
recorded draws at sample collection (RCCL over xGMI when the backend is "nccl").
"""
import torch


def chain_shard(rank, world, total_chains):
    """-> (global id of the first chain of `rank`, number of chains it owns); remainder to the low ranks."""
    base, rem = divmod(total_chains, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def gather_draws(dist, draws, gather_list, rank, dst=0):
    """Gathers every rank's [rows][P][chains_r] block to `dst` (equal shard sizes: one collective)."""
    dist.gather(draws, gather_list=gather_list if rank == dst else None, dst=dst)


def merge_gathered(blocks):
    """[rows][P][chains_r] per rank -> [rows][P][total chains], chains in global-id order."""
    return torch.cat(list(blocks), dim=2)
