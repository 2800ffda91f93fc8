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
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one-rank gather ok", flush=True)


if __name__ == "__main__":
    main()
