import numpy as np

from prosim_amd import synth
from prosim_amd.postprocess import replicate_scene, trajs_to_world
from prosim_amd.spec import SMALL_SPEC


def test_replicate_scene_shapes():
    s = synth.make_scene(SMALL_SPEC, 5, 7, batch=1, seed=0, goal=True)
    s["fut_obs_input"] = np.zeros((SMALL_SPEC.n_replans - 1, 1, 5, 11, 24), np.float32)
    r = replicate_scene(s, 4)
    assert r["obs_input"].shape[0] == 4 and r["cond"]["goal"]["input"].shape == (4, 5, 3)
    assert r["fut_obs_input"].shape[:2] == (SMALL_SPEC.n_replans - 1, 4)
    assert np.array_equal(r["map_pos"][3], s["map_pos"][0])


def test_world_frame_transform_round_trip():
    rng = np.random.RandomState(0)
    th = rng.uniform(-np.pi, np.pi, (3, 6))
    traj = np.zeros((3, 6, 4), np.float32)
    traj[..., 0], traj[..., 1] = rng.randn(3, 6), rng.randn(3, 6)
    traj[..., 2], traj[..., 3] = np.sin(th), np.cos(th)
    pos, h0 = rng.randn(3, 2).astype(np.float32) * 10, rng.uniform(-np.pi, np.pi, 3).astype(np.float32)
    w = trajs_to_world(traj, pos, h0)
    # inverse transform recovers the local trajectory
    d = w["xy"] - pos[:, None]
    c, s = np.cos(-h0)[:, None], np.sin(-h0)[:, None]
    back = np.stack([d[..., 0] * c - d[..., 1] * s, d[..., 0] * s + d[..., 1] * c], -1)
    assert np.abs(back - traj[..., :2]).max() < 1e-5
    assert np.abs(np.angle(np.exp(1j * (w["heading"] - h0[:, None] - th)))).max() < 1e-5
    assert w["heading"].min() >= -np.pi - 1e-6 and w["heading"].max() < np.pi + 1e-6
