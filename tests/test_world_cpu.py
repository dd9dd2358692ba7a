"""oracle/world_oracle.py against the reference's own obtain_rollout_trajs_in_world (tests/golden/ref_world_trajs.npz, made
by tests/gen_golden.py::gen_world), and the replica bookkeeping of the host side (no GPU)."""
import os

import numpy as np
import torch

from oracle import world_oracle as wo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_world_trajs.npz")


def test_world_oracle_matches_the_reference_bit_for_bit():
    g = np.load(GOLD)
    for s in (0, 1):
        got = wo.trajs_in_world(g[f"s{s}_traj"], g[f"s{s}_init_pos"], g[f"s{s}_init_head"], g[f"s{s}_tf"]).numpy()
        assert np.array_equal(got, g[f"s{s}_world"])
        # fp64 restatement: the fp32 result is within rounding of it (coordinates ~ 4 km: 1 ulp = 2.4e-4 m)
        g64 = wo.trajs_in_world(g[f"s{s}_traj"], g[f"s{s}_init_pos"], g[f"s{s}_init_head"], g[f"s{s}_tf"], dtype=torch.float64).numpy()
        assert np.abs(g64[..., :2] - got[..., :2]).max() < 2e-3
        assert np.abs(np.angle(np.exp(1j * (g64[..., 2] - got[..., 2])))).max() < 1e-5
        assert (got[..., 2] >= -np.pi).all() and (got[..., 2] < np.pi + 1e-6).all()


def test_identity_transform_and_round_trip():
    g = np.load(GOLD)
    traj, pos, th = g["s0_traj"], g["s0_init_pos"], g["s0_init_head"]
    c = wo.trajs_in_world(traj, pos, th, None, dtype=torch.float64).numpy()
    # undo: rotate back by -heading about the init position
    d = c[..., :2] - pos[:, None].astype(np.float64)
    cs, sn = np.cos(-th.astype(np.float64)), np.sin(-th.astype(np.float64))
    back = np.stack([d[..., 0] * cs - d[..., 1] * sn, d[..., 0] * sn + d[..., 1] * cs], -1)
    assert np.abs(back - traj[..., :2]).max() < 1e-9
    assert wo.replicate_rows(pos, 3).shape == (3 * len(pos), 2) and np.array_equal(wo.replicate_rows(pos, 3)[len(pos):2 * len(pos)], pos)
