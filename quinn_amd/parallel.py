"""Sharding of independent units (chains / ensemble members) over the ranks of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  Units are block-partitioned; the dataset is replicated; each unit carries its own
random stream (per-chain `RandomState(seed)`), so results do not depend on the number of ranks.
There is no collective on the data path: the only communication is ONE all_gather of the result
arrays at the end (what the API returns), padded to the largest shard.
"""
import numpy as np
import torch


def dist_info():
    """(rank, world) -- (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, rank=None, world=None):
    """Block partition of range(n): the [lo, hi) owned by `rank` (first n % world ranks get one more)."""
    if rank is None or world is None:
        rank, world = dist_info()
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _comm_device():
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_gather_rows(local, n_total):
    """Concatenate per-rank arrays along axis 0 on every rank.  `local`: numpy array holding this
    rank's shard_bounds(n_total) rows.  One all_gather (fixed-size, padded) -- no other traffic."""
    import torch.distributed as dist
    rank, world = dist_info()
    local = np.ascontiguousarray(local)
    if world == 1:
        return local
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    dev = _comm_device()
    pad = np.zeros((nmax,) + local.shape[1:], dtype=local.dtype)
    pad[:local.shape[0]] = local
    send = torch.as_tensor(pad, device=dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    parts = [recv[r][:hi - lo].cpu().numpy() for r, (lo, hi) in enumerate(sizes)]
    return np.concatenate(parts, axis=0)


def run_chains_sharded(make_sampler, nmcmc, param_ini, seeds, verbose=False):
    """Run len(seeds) chains split over the ranks; every rank returns the full result dict.

    Args:
        make_sampler: () -> a sampler with its log-posterior hooks set (MCMCBase subclass).
        param_ini: `(C,p)` initial states or None (then chain c starts at RandomState(seeds[c]).rand(p),
            the reference's nn_mcmc.py:124 on the chain's own stream; needs `pdim` on the sampler).
        seeds: C integers, one stream per chain.
    """
    C = len(seeds)
    lo, hi = shard_bounds(C)
    rngs = [np.random.RandomState(int(s)) for s in seeds[lo:hi]]
    mc = make_sampler()
    if param_ini is None:
        ini = np.stack([r.rand(mc.pdim) for r in rngs]) if rngs else np.zeros((0, mc.pdim))
    else:
        ini = np.asarray(param_ini, dtype=np.float64)[lo:hi]
    if hi > lo:
        res = mc.run(nmcmc, ini, rngs=rngs, verbose=verbose)
    else:
        p = ini.shape[1]
        res = {'chain': np.zeros((0, nmcmc + 1, p)), 'mapparams': np.zeros((0, p)), 'maxpost': np.zeros(0),
               'accrate': np.zeros(0), 'logpost': np.zeros((0, nmcmc + 1)), 'alphas': np.zeros((0, nmcmc + 1))}
    return {k: all_gather_rows(np.asarray(v), C) for k, v in res.items()}
