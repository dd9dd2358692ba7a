"""TEST INFRASTRUCTURE, data only: the synthetic scene dict (prosim_amd.synth's layout) as the ``batch.extras`` the reference reads
(dataset/format_utils.py:798-815) -- what a caller of ProSim.forward(batch, 'val') hands over, and therefore what the host-side mirror
of that interface (prosim_amd.modules.ProSimHip) is tested with.  No reference code, no stand-ins: this module runs anywhere (the -m gpu
tests import it; oracle/ref_harness.py, which also needs it, is for the build container only).  Never imported by the product."""
from __future__ import annotations

import numpy as np
import torch


class Extras:
    """Stands in for the trajdata batch object: only ``.extras`` is read on this path."""

    def __init__(self, extras):
        self.extras = extras


def make_batch(scene_in, spec):
    """scene_in (oracle layout, numpy) -> the ``batch.extras`` dict the reference reads
    (dataset/format_utils.py:798-815).  Observed agents = slots with a valid history point; policy agents =
    prompt_mask slots, handed to the reference as a DENSE prompt tensor in slot order with their agent ids (the
    reference matches prompts to observations by id, traj_sam.py:246-250)."""
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).clone()
    pm = scene_in["prompt_mask"].astype(bool)
    B, N = pm.shape
    seen0 = scene_in["obs_mask"].all(-1).any(-1)                      # [B, N] observed at the initial step
    assert (seen0 | ~pm).all(), "a policy agent must be observed"
    seen = seen0.copy()                                                # ... or in any later frame (agents that enter)
    if "fut_obs_mask" in scene_in:
        seen |= scene_in["fut_obs_mask"].all(-1).any(-1).any(0)
    n_obs = [int(seen[b].sum()) for b in range(B)]
    assert all(seen[b, :n_obs[b]].all() for b in range(B)), "observed agents must fill the leading slots"
    # every frame lists every agent that is ever observed; a frame's mask says whether the agent is in the scene at
    # that step (a listed agent without a valid point is no token: same as not listing it, obs_encoder.py:84-87)
    obs_ids = [[f"a{n}" for n in range(n_obs[b])] for b in range(B)]
    pol = [np.nonzero(pm[b])[0] for b in range(B)]
    Np = max(len(p_) for p_ in pol)

    def dense(arr, fill=0.0):                                          # [B, N, ...] -> [B, Np, ...] policy rows
        out = np.full((B, Np) + arr.shape[2:], fill, arr.dtype)
        for b in range(B):
            out[b, :len(pol[b])] = arr[b, pol[b]]
        return out

    pmask = np.zeros((B, Np), bool)
    for b in range(B):
        pmask[b, :len(pol[b])] = True
    pol_ids = [[f"a{n}" for n in pol[b]] for b in range(B)]
    slot2pol = np.full((B, N), 0, np.int64)
    for b in range(B):
        slot2pol[b, pol[b]] = np.arange(len(pol[b]))

    def obs(r=None):
        if r is None or "fut_obs_input" not in scene_in:
            return dict(input=t(scene_in["obs_input"]), mask=t(scene_in["obs_mask"], torch.bool),
                        position=t(scene_in["obs_pos"]), heading=t(scene_in["obs_head"]), agent_ids=obs_ids)
        return dict(input=t(scene_in["fut_obs_input"][r]), mask=t(scene_in["fut_obs_mask"][r], torch.bool),
                    position=t(scene_in["fut_obs_pos"][r]), heading=t(scene_in["fut_obs_head"][r]), agent_ids=obs_ids)

    later = [int(tt_) for tt_ in spec.all_t_indices if tt_ > 0]
    extras = dict(
        init_obs=obs(),
        init_map=dict(input=t(scene_in["map_input"]), mask=t(scene_in["map_mask"], torch.bool),
                      position=t(scene_in["map_pos"]), heading=t(scene_in["map_head"])),
        prompt=dict(motion_pred=dict(prompt=t(dense(scene_in["prompt"])), prompt_mask=t(pmask, torch.bool),
                                     position=t(dense(scene_in["obs_pos"])), heading=t(dense(scene_in["obs_head"]))[..., None],
                                     agent_type=t(dense(scene_in["agent_type"], 1), torch.long), agent_ids=pol_ids)),
        all_t_indices=torch.tensor(spec.all_t_indices),
        fut_obs={tt_: obs(i) for i, tt_ in enumerate(later)},
    )
    cond = {}
    for k, v in (scene_in.get("cond") or {}).items():
        # conditions index policy agents: slot indices -> rows of the dense prompt tensor; conditions of
        # non-policy slots are dropped (their mask is False)
        pi = v["prompt_idx"].astype(np.int64)                    # [B, C, 1] unary / [B, C, 2] binary (source, target)
        idx = np.stack([np.take_along_axis(slot2pol, pi[..., j], 1) for j in range(pi.shape[-1])], -1)
        cm = v["mask"].astype(bool)
        for j in range(pi.shape[-1]):
            cm = cm & np.take_along_axis(pm, pi[..., j], 1)
        cond[k] = dict(input=t(v["input"]), mask=t(cm, torch.bool), prompt_idx=t(idx, torch.long),
                       prompt_mask=t(pmask, torch.bool))
    extras["condition"] = cond
    return Extras(extras)
