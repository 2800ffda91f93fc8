"""CPU: the multi-rank path (world_size 2, gloo) -- sharded chains, one all_gather."""
import os
import subprocess
import sys
import tempfile

from quinn_amd.parallel import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_range():
    for n in (1, 5, 64, 513):
        for world in (1, 2, 3, 8):
            parts = [shard_bounds(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_chains_gloo():
    out = tempfile.mkdtemp(prefix="qn_dist_")
    env = dict(os.environ, QN_DIST_OUT=out, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert os.path.exists(os.path.join(out, "ok_0")) and os.path.exists(os.path.join(out, "ok_1"))
