"""N > 1 path on CPU: world_size-2 gloo processes exercise the env sharding and the observation all-gather."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from myosuite_mjx_amd.shard import shard_range, gather_obs, u01
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B, D = 8, 5
    lo, hi = shard_range(rank, world, B)
    assert (lo, hi) == (rank * B, (rank + 1) * B)
    # synthetic obs row = f(global env id): the gathered tensor must be in global env-id order on every rank
    obs = torch.tensor([[u01(1, e, d) for d in range(D)] for e in range(lo, hi)], dtype=torch.float32)
    g = gather_obs(dist, obs)
    ref = torch.tensor([[u01(1, e, d) for d in range(D)] for e in range(world * B)], dtype=torch.float32)
    assert g.shape == (world * B, D) and torch.equal(g, ref)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_two_rank_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", str(w)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
