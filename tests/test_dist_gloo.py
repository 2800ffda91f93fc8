"""CPU: the multi-rank path (world_size 2, gloo) -- sharded chains, one all_gather."""
import json
import os
import sys
import tempfile

from quinn_amd.parallel import launch_ranks, shard_bounds

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
    env = dict(os.environ, QN_DIST_OUT=out, OMP_NUM_THREADS="1")
    rc, so, se = launch_ranks(os.path.join(ROOT, "tests", "dist_worker.py"), [], 2, env=env, timeout=600, capture=True)
    assert rc == 0, so[-2000:] + se[-4000:]
    assert os.path.exists(os.path.join(out, "ok_0")) and os.path.exists(os.path.join(out, "ok_1"))


def test_launcher_starts_the_ranks_and_forwards_their_output():
    """`launch_ranks` (what `bench.py --gpus N` uses): N children with RANK / WORLD_SIZE set, rendezvous on 127.0.0.1,
    exit code and output forwarded."""
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, "w.py")
        with open(w, "w") as f:
            f.write("import os, sys, json\nimport torch.distributed as dist\ndist.init_process_group('gloo')\n"
                    "import torch\nt = torch.tensor([float(dist.get_rank() + 1)])\ndist.all_reduce(t)\n"
                    "if dist.get_rank() == 0: print(json.dumps({'world': dist.get_world_size(), 'sum': t.item(), 'arg': sys.argv[1]}))\n"
                    "dist.destroy_process_group()\nsys.exit(3 if sys.argv[1] == 'fail' else 0)\n")
        rc, so, se = launch_ranks(w, ["hello"], 2, timeout=300, capture=True)
        assert rc == 0, se[-3000:]
        line = [ln for ln in so.splitlines() if ln.startswith("{")][-1]
        assert json.loads(line) == {"world": 2, "sum": 3.0, "arg": "hello"}
        rc, _, _ = launch_ranks(w, ["fail"], 2, timeout=300, capture=True)
        assert rc != 0
