"""Worker for tests/test_gpu_00_launch.py: ONE rank with an RCCL ("nccl") process group on cuda:0; `gather_rows` goes through its
collectives (QN_FORCE_GATHER=1) on device tensors -- all_gather_into_tensor / gather in bounded pieces, float64 / int64 / int32 /
float32, odd piece sizes -- and a device AMCMC run gathers its result dict.  Everything of the N > 1 result path but the transport."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from quinn_amd.parallel import gather_results, gather_rows  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    saved = os.dup(1)
    os.dup2(2, 1)                                            # (RCCL's banner: not on this worker's stdout)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        probe = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(probe)
        torch.cuda.synchronize(dev)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    assert float(probe.item()) == 1.0
    rs = np.random.RandomState(0)
    for dt in (torch.float64, torch.float32, torch.int64, torch.int32):
        a = torch.as_tensor(rs.randint(-1000, 1000, size=(7, 5, 3))).to(dt).to(dev)
        for dst in ("all", "root"):
            for chunk in (8, 56, 1 << 20):
                out = gather_rows(a, 7, dst=dst, chunk_bytes=chunk)
                assert out.dtype == a.cpu().numpy().dtype and np.array_equal(out, a.cpu().numpy()), (dt, dst, chunk)
    host = rs.randn(4, 9)                                    # a host array goes through the device piece buffers as well
    assert np.array_equal(gather_rows(host, 4, dst="all", chunk_bytes=40), host)
    # a sampler's result dict, device tensors
    res = {"chain": torch.randn(3, 11, 6, device=dev, dtype=torch.float64), "mapparams": torch.randn(3, 6, device=dev, dtype=torch.float64),
           "maxpost": torch.randn(3, device=dev, dtype=torch.float64), "accrate": torch.rand(3, device=dev, dtype=torch.float64),
           "logpost": torch.randn(3, 11, device=dev, dtype=torch.float64), "alphas": torch.rand(3, 11, device=dev, dtype=torch.float64)}
    got = gather_results(res, 3, gather="all", gather_chain="root")
    for k, v in res.items():
        assert np.array_equal(got[k], v.cpu().numpy()), k
    # the solver path: NN_MCMC.fit(engine='device') gathers its result dict from the device buffers -- through RCCL (forced) and by
    # the single-process short cut: same chains, bit for bit
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_mcmc import NN_MCMC
    torch.set_default_dtype(torch.double)
    x = rs.rand(200, 1) * 4 - 2
    y = np.sin(2 * x) + 0.1 * rs.randn(200, 1)

    def fit(sampler, gather):
        torch.manual_seed(0)
        uq = NN_MCMC(MLP(1, 1, (16, 16), activ='tanh'), verbose=False)
        if sampler == "amcmc":
            uq.fit(x, y, zflag=False, datanoise=0.1, nmcmc=300, sampler='amcmc', sampler_params={'gamma': 0.1, 't0': 50, 'tadapt': 100},
                   seeds=range(5), engine='device', gather=gather)
        else:
            uq.fit(x, y, zflag=False, datanoise=0.1, nmcmc=40, sampler='hmc', sampler_params={'epsilon': 0.002, 'L': 3},
                   seeds=range(5), engine='device', gather=gather)
        return uq.mcmc_results

    for sampler in ("amcmc", "hmc"):
        for gather in ("all", "root"):
            os.environ["QN_FORCE_GATHER"] = "1"
            a = fit(sampler, gather)
            os.environ["QN_FORCE_GATHER"] = "0"
            b = fit(sampler, gather)
            for k in a:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (sampler, gather, k)
            assert np.isfinite(a["logpost"]).all() and a["chain"].shape[0] == 5
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one-rank gather ok", flush=True)


if __name__ == "__main__":
    main()
