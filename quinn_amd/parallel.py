"""Sharding of independent units (chains / ensemble members) over the ranks of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  Units are block-partitioned; the dataset is replicated; each unit carries its own
random stream (per-chain `RandomState(seed)` on the host engines, Philox keyed by the GLOBAL chain
id on the device engines), so results do not depend on the number of ranks.  There is no
collective on the data path: the only communication is ONE gather of the result arrays at the end
(`gather_rows`), taken straight from the device buffers in bounded chunks.

`launch_ranks` starts the N ranks of a script (`python -m torch.distributed.run`) as a CHILD
process; it must be called before the calling process has touched the GPU.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import torch

GATHER_MODES = ("all", "root", "none")
# a gather whose result would exceed this many bytes on a receiving rank raises instead of exhausting host memory
DEFAULT_MAX_GATHER_BYTES = 64 << 30
DEFAULT_CHUNK_BYTES = 64 << 20


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


def gather_bytes(shape_tail, itemsize, n_total, world=None, dst="all"):
    """Bytes moved by `gather_rows` of an `[n_total, *shape_tail]` array: (sent per rank, received per receiving
    rank, host bytes of the result on a receiving rank).  Padding to the largest shard included."""
    if world is None:
        world = dist_info()[1]
    row = int(np.prod(shape_tail, dtype=np.int64)) * itemsize
    nmax = -(-n_total // world)
    if dst == "none" or world == 1:
        return 0, 0, 0
    return nmax * row, world * nmax * row, n_total * row


def gather_rows(local, n_total, dst="all", chunk_bytes=DEFAULT_CHUNK_BYTES, max_bytes=DEFAULT_MAX_GATHER_BYTES):
    """Concatenate the per-rank shards (`shard_bounds(n_total)` rows each) along axis 0.

    local: this rank's rows -- a torch tensor (device or host) or a numpy array.  Device tensors are sent from
        where they are in pieces of at most `chunk_bytes` per rank (no whole-array host copy, no re-upload);
        the receive buffer is `world * chunk_bytes`, copied piece by piece into the host result.
    dst: 'all' -- every rank returns the full `[n_total, ...]` numpy array (one all_gather per piece);
         'root' -- rank 0 returns the full array, the others their own shard (one gather per piece);
         'none' -- no communication, every rank returns its own shard.
    Raises MemoryError (on every rank, before any traffic) if the result would exceed `max_bytes` on a
    receiving rank: gather less (`dst='root'` / `'none'`, thin the chain) or raise the limit.
    """
    import torch.distributed as dist
    if dst not in GATHER_MODES:
        raise ValueError(f"gather mode {dst!r} is not one of {GATHER_MODES}")
    rank, world = dist_info()
    t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(local))
    # (QN_FORCE_GATHER=1: a single rank with a live process group goes through the collectives too -- the one-GPU rehearsal of the
    # RCCL path, tests/test_gpu_00_launch.py)
    forced = os.environ.get("QN_FORCE_GATHER") == "1" and dist.is_available() and dist.is_initialized()
    if (world == 1 and not forced) or dst == "none":
        return t.detach().cpu().numpy()
    lo, hi = shard_bounds(n_total, rank, world)
    if t.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {t.shape[0]} rows, its shard of {n_total} is {hi - lo}")
    tail = tuple(t.shape[1:])
    row = int(np.prod(tail, dtype=np.int64))
    total_bytes = n_total * row * t.element_size()
    if max_bytes is not None and total_bytes > max_bytes:
        raise MemoryError(f"gathering [{n_total}, {', '.join(map(str, tail))}] {t.dtype} = {total_bytes / 2**30:.1f} GiB "
                          f"to {'every rank' if dst == 'all' else 'rank 0'} exceeds max_bytes = {max_bytes / 2**30:.1f} GiB: "
                          "use gather='root' / 'none', thin the chain, or raise the limit")
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    nmax = max(b - a for a, b in sizes)
    receiver = dst == "all" or rank == 0
    out = np.empty((n_total,) + tail, dtype=torch.empty(0, dtype=t.dtype).numpy().dtype) if receiver else None
    out_flat = out.reshape(-1) if receiver else None
    cdev = _comm_device()
    flat = t.detach().contiguous().reshape(-1)
    nloc, nflat = flat.numel(), nmax * row
    piece = max(1, int(chunk_bytes) // max(1, t.element_size()))
    send = torch.zeros(min(piece, max(nflat, 1)), dtype=t.dtype, device=cdev)
    recv = torch.empty(world, send.numel(), dtype=t.dtype, device=cdev) if receiver else None
    for s in range(0, nflat, piece):
        e = min(nflat, s + piece)
        n_valid = max(0, min(e, nloc) - s)
        if n_valid < e - s:
            send[:e - s].zero_()                              # padding of a short (or empty) shard
        if n_valid > 0:
            send[:n_valid].copy_(flat[s:s + n_valid])         # device->device (nccl) or device->host (gloo) piece
        if dst == "all":
            dist.all_gather_into_tensor(recv.reshape(-1), send)
        else:
            dist.gather(send, list(recv.unbind(0)) if rank == 0 else None, dst=0)
        if receiver:
            host = recv.cpu().numpy()
            for r, (a, b) in enumerate(sizes):
                nv = max(0, min(e, (b - a) * row) - s)
                if nv > 0:
                    out_flat[a * row + s:a * row + s + nv] = host[r, :nv]
    if receiver:
        return out
    return t.detach().cpu().numpy()


def all_gather_rows(local, n_total):
    """`gather_rows(..., dst='all')` (kept for callers of the round-1 name)."""
    return gather_rows(local, n_total, dst="all")


def gather_results(res, n_total, gather="all", gather_chain=None, max_bytes=DEFAULT_MAX_GATHER_BYTES):
    """Gather a sampler's result dict (`chain`, `mapparams`, `maxpost`, `accrate`, `logpost`, `alphas`; device
    tensors or numpy, this rank's shard) -> dict of numpy arrays.  The small entries follow `gather`; `chain`
    ([C, nmcmc+1, p], the large one: 43.6 GB per rank at cfg2) follows `gather_chain` (None: the same as `gather`)."""
    gc = gather if gather_chain is None else gather_chain
    return {k: (None if v is None else gather_rows(v, n_total, dst=gc if k == "chain" else gather, max_bytes=max_bytes))
            for k, v in res.items()}


def run_chains_sharded(make_sampler, nmcmc, param_ini, seeds, verbose=False, gather="all", gather_chain=None):
    """Run len(seeds) chains split over the ranks; the result dict follows `gather` (see `gather_rows`).

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
        res = empty_results(nmcmc, ini.shape[1])
    return gather_results({k: np.asarray(v) for k, v in res.items()}, C, gather, gather_chain)


def empty_results(nmcmc, p):
    """Result dict of a rank that owns no chain (fewer chains than ranks)."""
    return {'chain': np.zeros((0, nmcmc + 1, p)), 'mapparams': np.zeros((0, p)), 'maxpost': np.zeros(0),
            'accrate': np.zeros(0), 'logpost': np.zeros((0, nmcmc + 1)), 'alphas': np.zeros((0, nmcmc + 1))}


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(script, script_args, nproc, env=None, timeout=None, port=None, capture=False):
    """Start `nproc` ranks of `script` on this node as a child `python -m torch.distributed.run` (rendezvous on
    127.0.0.1) and wait for it.  Call this BEFORE the calling process has made any GPU call: a process that
    has initialised the GPU must neither fork GPU children nor replace itself.

    Returns the child's exit code (capture=False: its output goes to this process's stdout / stderr) or
    `(exit code, stdout, stderr)` (capture=True)."""
    if torch.cuda.is_initialized():
        raise RuntimeError("launch_ranks must run before this process touches the GPU")
    env = dict(os.environ if env is None else env)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL between processes on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + [str(a) for a in script_args]
    r = subprocess.run(cmd, env=env, timeout=timeout, capture_output=capture, text=capture)
    if capture:
        return r.returncode, r.stdout, r.stderr
    return r.returncode
