"""SURVEY section 8 (f2) on the device: M replicas of one scene that share encode_scene + generate_policy
(ps_set_replicas; parallel_rollout_batch / replica_batch_for_parallel_rollout, rollout/gpu_utils.py:59-123, :179-228) and
the world-frame output kernel (ps_world_trajs; obtain_rollout_trajs_in_world :230-281) against the reference-made fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import prosim_oracle as orc, world_oracle as wo
from prosim_amd import synth, weights
from prosim_amd.engine import Engine
from prosim_amd.postprocess import replicate_scene
from prosim_amd.spec import SMALL_SPEC
from golden_cases import SPECS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_world_trajs_kernel_vs_reference_fixture():
    g = np.load(os.path.join(GOLD, "ref_world_trajs.npz"))
    spec = SMALL_SPEC
    eng = Engine(spec, weights.init_weights(spec, 0))
    try:
        for s in (0, 1):
            traj, pos, th, tf = g[f"s{s}_traj"], g[f"s{s}_init_pos"], g[f"s{s}_init_head"], g[f"s{s}_tf"]
            n, T = traj.shape[:2]
            assert T == spec.max_steps
            scene = synth.make_scene(spec, n, 8, batch=1, seed=3)
            scene["obs_pos"], scene["obs_head"] = pos[None].copy(), th[None, :, 0].copy()
            eng.set_scene(scene)
            assert eng.num_agents == n
            H = spec.hist_steps
            full = np.zeros((n, H + T, 4), np.float32)
            full[:, H:] = traj
            eng.set_state(full, np.zeros((n, H + T, 2), np.float32))
            got = eng.world_trajs(tf)
            want = g[f"s{s}_world"]
            # fp32 on both sides; device sin / cos / atan2 differ from libm by an ulp: 1 ulp of a 4 km coordinate = 2.4e-4 m
            assert err(got[..., :2], want[..., :2]) < 1e-3
            assert np.abs(np.angle(np.exp(1j * (got[..., 2].astype(np.float64) - want[..., 2])))).max() < 2e-6
            # identity transform = the scene-centre frame; and into a caller-owned device buffer
            out = torch.zeros(n, T, 3, device="cuda")
            eng.world_trajs(None, out.data_ptr())
            eng.sync()
            c64 = wo.trajs_in_world(traj, pos, th, None, dtype=torch.float64).numpy()
            assert err(out.cpu().numpy()[..., :2], c64[..., :2]) < 1e-4
    finally:
        eng.close()


@pytest.mark.parametrize("case", ["k1", "k3_modes", "replay"])
def test_replicas_share_the_prefix_and_match_the_replicated_batch(case):
    """ps_set_replicas(M) on ONE scene == the reference's way (the M-fold replicated batch through the whole path), and ==
    the fp64 oracle of that batch; with K = 3 modes every replica follows its own draw."""
    M = 5
    spec = SPECS["small_k3"] if case == "k3_modes" else SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 23, 70, batch=1, seed=11, goal=(case == "k1"), tags=(case == "k1"), replay=(0.25 if case == "replay" else 0.0))
    rng = np.random.default_rng(5)
    N = scene["prompt_mask"].shape[1]
    choice = rng.integers(0, spec.motion_k, (spec.n_replans, M, N)).astype(np.int32) if case == "k3_modes" else None
    tiled = replicate_scene(scene, M)
    if choice is not None:
        tiled["mode_choice"] = choice
    with torch.no_grad():
        o64 = orc.rollout(w, spec, tiled, dtype=torch.float64)
        o32 = orc.rollout(w, spec, tiled, dtype=torch.float32)
    # what fp32 arithmetic alone does to this closed loop (the K = 3 heads amplify more than the K = 1 head)
    floor = float((o32["traj"].double() - o64["traj"]).abs().max())
    a, b = Engine(spec, w), Engine(spec, w)
    try:
        a.set_replicas(M)
        a.set_scene(dict(scene, mode_choice=choice) if choice is not None else scene)
        b.set_scene(tiled)
        a.rollout(); b.rollout()
        A = a.num_agents
        assert A == b.num_agents and a.num_map_tokens * M == b.num_map_tokens        # the map exists once
        assert np.array_equal(a.row_slots, b.row_slots) and a.padded("traj").shape == b.padded("traj").shape
        # replica 0 of the prefix == every replica of it
        emd = a.get("policy_emd").reshape(M, A // M, -1)
        assert all(np.array_equal(emd[0], emd[r]) for r in range(1, M))
        tok = a.get("scene_tokens")[a.num_map_tokens:].reshape(M, A // M, -1)
        assert tok.shape[1] == A // M
        for name in ("policy_emd", "motion_pred", "traj"):
            assert err(a.get(name), b.get(name)) < (2e-5 if name != "traj" else 5e-4), name
        pol = a.policy_rows                                                    # the oracle's pairs are the policy agents
        assert err(a.get("motion_pred")[0][pol], o64["motion_pred"][:int(pol.sum())].numpy()) < TOL
        pm = tiled["prompt_mask"].astype(bool)
        d = np.abs(a.padded("traj") - o64["traj"].numpy())[pm].reshape(int(pm.sum()), -1).max(1)
        assert d.max() < 3 * floor + TOL and np.median(d) < floor + TOL, (d.max(), floor)
        if case == "k3_modes":   # the replicas really differ
            t = a.padded("traj")
            assert err(t[0], t[1]) > 1e-3
        else:                    # ... and without a draw they are the same rollout M times
            t = a.padded("traj")
            assert all(np.array_equal(t[0], t[r]) for r in range(1, M))
        # switching the mode off again restores the plain batch behaviour
        a.set_replicas(1)
        a.set_scene(tiled)
        a.rollout()
        assert np.array_equal(a.get("traj"), b.get("traj"))
    finally:
        a.close(); b.close()


def test_replicas_argument_checks():
    spec = SMALL_SPEC
    eng = Engine(spec, weights.init_weights(spec, 0))
    try:
        eng.set_replicas(3)
        with pytest.raises(RuntimeError, match="ONE scene"):
            eng.set_scene(synth.make_scene(spec, 6, 12, batch=2, seed=1))
        with pytest.raises(RuntimeError, match="replicas >= 1"):
            eng.set_replicas(0)
        with pytest.raises(RuntimeError, match="before a rollout"):
            eng.world_trajs()
    finally:
        eng.close()
