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


def gather_draws(dist, draws, gather_list, rank, dst=0, equal_sizes=None):
    """Gathers every rank's [rows][P][chains_r] block to `dst`: one gather collective when all shards have the same size (the
    bench), point-to-point send/recv when they differ (total chains not a multiple of the world size)."""
    world = dist.get_world_size()
    if equal_sizes is None:     # callers that know (bench.py: the same chain count on every rank) skip this small collective
        sizes = torch.zeros(world, dtype=torch.int64, device=draws.device)
        sizes[rank] = draws.shape[2]
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
        equal_sizes = bool((sizes == sizes[0]).all())
    if equal_sizes:
        dist.gather(draws, gather_list=gather_list if rank == dst else None, dst=dst)
        return
    if rank == dst:
        gather_list[dst].copy_(draws)
        for r in range(world):
            if r != dst:
                dist.recv(gather_list[r], src=r)
    else:
        dist.send(draws.contiguous(), dst=dst)


def merge_gathered(blocks):
    """[rows][P][chains_r] per rank -> [rows][P][total chains], chains in global-id order."""
    return torch.cat(list(blocks), dim=2)


def pooled_moments(dist, draws):
    """Posterior mean / sd per recorded component over the draws of ALL ranks ([rows][P][chains_r] per rank): the two-pass
    all-reduce of csrc/amwg_group.hip (amwg_group_moments) for the one-process-per-GPU layout -- pass 1 all-reduces
    (sum, count), pass 2 the squared deviations from the global mean.  Works on CPU tensors (gloo) and device tensors (RCCL)."""
    n = torch.tensor([float(draws.shape[0] * draws.shape[2])], dtype=torch.float64, device=draws.device)
    acc = torch.cat([draws.sum(dim=(0, 2)), n])
    if dist is not None:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    total = acc[-1]
    mean = acc[:-1] / total
    ss = ((draws - mean.view(1, -1, 1)) ** 2).sum(dim=(0, 2))
    if dist is not None:
        dist.all_reduce(ss, op=dist.ReduceOp.SUM)
    sd = torch.sqrt(ss / (total - 1)) if float(total) > 1 else torch.zeros_like(ss)
    return mean, sd


def pooled_convergence(dist, draws):
    """Split-R-hat and ESS over the chains of ALL ranks (same definitions as amwg_group_diagnostics): per-chain half means and
    variances locally, two small all-reduces for the sums across ranks."""
    rows, P, C = draws.shape
    half = rows // 2
    h = [draws[:half], draws[half:2 * half]]
    m = torch.stack([x.mean(dim=0) for x in h])                    # [2][P][C]
    v = torch.stack([x.var(dim=0, unbiased=True) for x in h])
    cm = 0.5 * (m[0] + m[1])
    a = torch.cat([v.sum(dim=(0, 2)), m.sum(dim=(0, 2)), cm.sum(dim=1), torch.tensor([float(C)], dtype=torch.float64, device=draws.device)])
    if dist is not None:
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
    Ct = a[-1]
    mh = 2.0 * Ct
    W, gm, gmc = a[:P] / mh, a[P:2 * P] / mh, a[2 * P:3 * P] / Ct
    b = torch.cat([((m - gm.view(1, -1, 1)) ** 2).sum(dim=(0, 2)), ((cm - gmc.view(-1, 1)) ** 2).sum(dim=1)])
    if dist is not None:
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
    var_plus = (half - 1) / half * W + b[:P] / (mh - 1)
    rhat = torch.sqrt(var_plus / W)
    ess = Ct * var_plus / (b[P:] / (Ct - 1))
    return rhat, ess
