"""world_size-2 gloo test of the scene sharding + metric gather (the N > 1 path of bench.py),
run as real processes through torch.distributed.run on 127.0.0.1 (no GPU needed)."""
import os
import subprocess
import sys
import textwrap

import torch

from prosim_amd.distributed import gather_scene_metrics, reduce_metrics, shard_scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_scenes_partition():
    for n, world in ((64, 8), (5, 2), (3, 4), (0, 2)):
        parts = [shard_scenes(n, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(n))
        assert all(i % world == r for r, p in enumerate(parts) for i in p)      # rollout/callbacks.py:76


def test_single_process_gather_is_identity():
    local = torch.arange(2 * 3 * 2, dtype=torch.float32).reshape(2, 3, 2)
    out = gather_scene_metrics(local, [0, 1], 2, 3)
    assert torch.equal(out, local)
    assert reduce_metrics(out)["scenes"] == 2


WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, ROOT_PLACEHOLDER)
    from prosim_amd.distributed import shard_scenes, gather_scene_metrics, reduce_metrics
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_scenes, max_agents = 5, 4                       # uneven shards: rank 0 owns 3 scenes, rank 1 owns 2
    mine = shard_scenes(n_scenes, rank, world)
    local = torch.full((len(mine), max_agents, 2), float("nan"))
    for j, s in enumerate(mine):
        n_ag = 1 + s % max_agents                     # ragged agent counts
        local[j, :n_ag, 0] = 10.0 * s + torch.arange(n_ag)
        local[j, :n_ag, 1] = 100.0 * s
    out = gather_scene_metrics(local, mine, n_scenes, max_agents)
    for s in range(n_scenes):
        n_ag = 1 + s % max_agents
        assert torch.equal(out[s, :n_ag, 0], 10.0 * s + torch.arange(n_ag)), (rank, s, out[s])
        assert torch.isnan(out[s, n_ag:]).all()
    # the reusable form bench.py keeps across rollouts: buffers and the scene index are built once, later calls
    # only move data (fresh values every call, padding rows stay NaN)
    from prosim_amd.distributed import SceneMetricGather
    g = SceneMetricGather(mine, n_scenes, max_agents, 2, "cpu")
    for it in range(3):
        again = g(local + 1000.0 * it)
        for s in range(n_scenes):
            n_ag = 1 + s % max_agents
            assert torch.equal(again[s, :n_ag, 0], 10.0 * s + torch.arange(n_ag) + 1000.0 * it), (rank, it, s)
            assert torch.isnan(again[s, n_ag:]).all()
    red = reduce_metrics(out)
    exp_ade = sum((10.0 * s + (1 + s % max_agents - 1) / 2.0) for s in range(n_scenes)) / n_scenes
    assert abs(red["rollout_ade"] - exp_ade) < 1e-5 and red["scenes"] == n_scenes, red
    dist.barrier()
    if rank == 0:
        print("GLOO_OK", red)
    dist.destroy_process_group()
""")


def test_world2_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.replace('ROOT_PLACEHOLDER', repr(ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GLOO_OK" in r.stdout
