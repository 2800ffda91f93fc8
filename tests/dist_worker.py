"""Worker for tests/test_dist_gloo.py: launched by torch.distributed.run with world_size 2 on the
CPU (gloo).  Exercises the N>1 path: chains sharded over ranks, per-chain random streams, ONE
all_gather at the end; results must equal the single-process run and the reference fixture."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden, spec_of, assert_chain_matches_fixture  # noqa: E402
from oracle import mlp_ref  # noqa: E402
from quinn_amd.mcmc.admcmc import AMCMC  # noqa: E402
from quinn_amd.parallel import all_gather_rows, gather_bytes, gather_rows, run_chains_sharded, shard_bounds  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(1)
    # 1. uneven gather
    n = 5
    lo, hi = shard_bounds(n)
    local = np.arange(lo, hi, dtype=np.float64)[:, None] * np.ones((1, 3))
    full = all_gather_rows(local, n)
    assert full.shape == (5, 3) and np.array_equal(full[:, 0], np.arange(5.0)), full
    assert [shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    # 1b. the same from a torch tensor, in pieces much smaller than a shard (7 elements per piece), every gather mode
    big = torch.arange(lo * 11, hi * 11, dtype=torch.float64).reshape(hi - lo, 11)
    want = np.arange(5 * 11, dtype=np.float64).reshape(5, 11)
    assert np.array_equal(gather_rows(big, n, dst="all", chunk_bytes=56), want)
    at_root = gather_rows(big, n, dst="root", chunk_bytes=56)
    assert np.array_equal(at_root, want if rank == 0 else want[lo:hi])
    assert np.array_equal(gather_rows(big, n, dst="none"), want[lo:hi])
    ints = gather_rows(torch.arange(lo, hi, dtype=torch.int64), n, dst="all", chunk_bytes=8)
    assert ints.dtype == np.int64 and np.array_equal(ints, np.arange(5))
    try:                                                    # the size guard trips on every rank before any traffic
        gather_rows(big, n, dst="all", max_bytes=100)
        raise AssertionError("no MemoryError")
    except MemoryError as err:
        assert "gather='root'" in str(err)
    assert gather_bytes((11,), 8, 5, world=2, dst="all") == (3 * 88, 2 * 3 * 88, 5 * 88)
    # 2. sharded chains == fixture == single-process lock-step
    g = load_golden("g8_multichain.npz")
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    yd = [v for v in g["y"]]
    lp = lambda w: mlp_ref.logpost(mod, w, g["x"], yd, float(g["sigma"]))
    C = int(g["nchains"])
    seeds = [int(g["seed0"]) + c for c in range(C)]

    def make():
        mc = AMCMC(gamma=float(g["gamma"]), t0=int(g["t0"]), tadapt=int(g["tadapt"]))
        mc.setLogPost(lp, None)
        mc.pdim = spec.nparams
        return mc
    res = run_chains_sharded(make, int(g["nmcmc"]), None, seeds)
    assert res["chain"].shape == g["chain"].shape
    for c in range(C):
        assert_chain_matches_fixture(res, g, c)
    if rank == 0:
        rngs = [np.random.RandomState(s) for s in seeds]
        ini = np.stack([r.rand(spec.nparams) for r in rngs])
        single = make().run(int(g["nmcmc"]), ini, rngs=rngs, verbose=False)
        for k in ("chain", "logpost", "alphas", "accrate", "mapparams", "maxpost"):
            assert np.array_equal(res[k], single[k], equal_nan=True), k
    # 3. fewer units than ranks: rank 1 owns an empty shard and still takes part in the gather
    lo, hi = shard_bounds(1)
    one = all_gather_rows(np.full((hi - lo, 2), 7.0), 1)
    assert one.shape == (1, 2) and (one == 7.0).all()
    res1 = run_chains_sharded(make, 25, None, seeds[:1])
    assert res1["chain"].shape == (1, 26, spec.nparams)
    # 4. gather='root': rank 0 holds every chain, rank 1 its own shard; 'none': shards only
    resr = run_chains_sharded(make, 25, None, seeds[:3], gather="root")
    assert resr["chain"].shape[0] == (3 if rank == 0 else 1)
    resn = run_chains_sharded(make, 25, None, seeds[:3], gather="none")
    lo3, hi3 = shard_bounds(3)
    assert resn["chain"].shape[0] == hi3 - lo3
    if rank == 0:
        assert np.array_equal(resr["chain"][lo3:hi3], resn["chain"])
    # small entries everywhere, the chain left sharded
    resm = run_chains_sharded(make, 25, None, seeds[:3], gather="all", gather_chain="none")
    assert resm["chain"].shape[0] == hi3 - lo3 and resm["logpost"].shape == (3, 26) and resm["mapparams"].shape[0] == 3
    assert np.array_equal(resm["chain"], resn["chain"])
    np.testing.assert_allclose(res1["chain"][0], g["chain"][0][:26], rtol=1e-9, atol=1e-11)
    dist.barrier()
    with open(os.path.join(os.environ["QN_DIST_OUT"], f"ok_{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
