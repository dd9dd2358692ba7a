"""GPU tests of the host-side mirror of the reference plugin interface (prosim_amd/modules.py):
same registry names, same call signatures, same output dict layout as prosim/models/traj_sam.py."""
import os

import numpy as np
import pytest
import torch

from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from oracle import prosim_oracle as orc
from oracle.ref_batch import make_batch
from golden_cases import FULL_CASES, SPECS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.fixture(scope="module")
def model():
    from prosim_amd import modules
    cls = modules.registry.get_model("prosim_policy_relpe_T_step_temporal_close_loop")   # MODEL.TYPE (trainer.py:160)
    m = cls(SMALL_SPEC, weights.init_weights(SMALL_SPEC, 0)).eval()
    yield m
    m.engine.close()


def test_registry_names_match_reference():
    from prosim_amd import modules
    r = modules.registry
    assert r.get_scene_encoder("attn_fusion_relpe") is modules.HipSceneEncoder      # attn_fusion.py:11
    assert r.get_decoder("attn_fusion_relpe") is modules.HipDecoder                # sym_coord.py:15
    assert r.get_policy("rel_pe_temporal") is modules.HipPolicy                    # policy/base.py:9
    assert r.get_policy("nope") is None


@pytest.mark.parametrize("name", ["small_ragged_b2", "small_drag_b2"])
def test_forward_output_contract_vs_fixture(model, name):
    """ProSim.forward(batch,'val') drop-in: same keys/shapes as traj_sam.py:562-595 and the values of
    the reference fixture (reference Python + stand-ins); small_drag_b2 carries all three condition types of the
    demo config (goal, v_action_tag, drag_point) through ``batch.extras['condition']``."""
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    batch = make_batch(scene, spec)
    out = model(batch, "val")["motion_pred"]
    assert set(out) >= {"motion_pred", "motion_prob", "pair_names", "reconst_pred", "rollout_trajs"}
    A = int(scene["prompt_mask"].sum())
    assert out["motion_pred"].shape == (spec.n_replans * A, 1, 10, 5) and len(out["pair_names"]) == spec.n_replans * A
    assert out["pair_names"][0] == "0-a0-0" and out["pair_names"][-1].endswith("-70")
    assert err(out["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < 1e-4
    floor = g["fp32_floor"][0]
    for b in range(2):
        for n in range(int(scene["prompt_mask"][b].sum())):
            r = out["rollout_trajs"][f"{b}-a{n}"]
            assert r["traj"].shape == (80, 4) and r["vel"].shape == (80, 2) and r["init_pos"].shape == (2,) and r["init_heading"].shape == (1,)
            assert err(r["traj"].numpy(), g["traj"][b, n]) < 3 * floor + 1e-4
    with pytest.raises(NotImplementedError):
        model(batch, "train")


def test_forward_with_log_replay_agents_vs_fixture(model):
    """Policy agents are a subset of the observed agents (matched by id, traj_sam.py:246-250); the others replay
    ``fut_obs``.  Outputs come back for the policy agents only, in prompt order, with the reference's values."""
    name = "small_replay_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    batch = make_batch(scene, spec)
    out = model(batch, "val")["motion_pred"]
    pm = scene["prompt_mask"].astype(bool)
    A = int(pm.sum())
    assert A < int(scene["obs_mask"].all(-1).any(-1).sum())                      # the case really has replay agents
    assert model.engine.num_policy_agents == A
    assert out["motion_pred"].shape == (spec.n_replans * A, 1, 10, 5) and g["motion_pred"].shape[0] == spec.n_replans * A
    assert err(out["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < 1e-4
    assert err(out["reconst_pred"][:A].numpy(), g["reconst_pred"]) < 1e-4
    floor = g["fp32_floor"][0]
    names = set()
    for b in range(2):
        for n in np.nonzero(pm[b])[0]:
            r = out["rollout_trajs"][f"{b}-a{n}"]
            names.add(f"{b}-a{n}")
            assert err(r["traj"].numpy(), g["traj"][b, n]) < 3 * floor + 1e-4
    assert set(out["rollout_trajs"]) == names                                     # and nothing for the replay agents
    # a policy agent that is not observed is an error, not a silent drop
    bad = make_batch(scene, spec)
    bad.extras["prompt"]["motion_pred"]["agent_ids"][0][0] = "ghost"
    with pytest.raises(ValueError, match="not among the observed"):
        model(bad, "val")


@pytest.mark.parametrize("name", ["small_replay_b2", "small_drag_b2"])
def test_staged_model_methods_equal_forward(model, name):
    """encode_scene -> encode_prompt -> generate_policy -> init_agent_trajs -> rollout_batch, the reference's own
    call sequence (traj_sam.py:59-116, :144-176), gives the values of the one-graph forward() bit for bit --
    including a batch whose policy agents are a subset of the observed agents."""
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    scene = synth.make_scene(spec, **kw)
    batch = make_batch(scene, spec)
    whole = model(batch, "val")["motion_pred"]
    scene_embs = model.encode_scene(batch)
    assert scene_embs["scene_tokens"].shape[0] == scene_embs["scene_type"].numel()
    n_obs = int(scene["obs_mask"].all(-1).any(-1).sum())
    assert int((scene_embs["scene_type"] == 1).sum()) == n_obs                 # every observed agent is a token
    prompt_encs = model.encode_prompt(batch)
    policy_emds = model.generate_policy(batch, scene_embs, prompt_encs)
    dm = batch.extras["prompt"]["motion_pred"]["prompt_mask"]
    assert policy_emds["motion_pred"]["emd"].shape[:2] == dm.shape
    ids = {"motion_pred": batch.extras["prompt"]["motion_pred"]["agent_ids"]}
    trajs = model.init_agent_trajs(ids, batch)
    out = model.rollout_batch(batch, scene_embs, policy_emds, ids, trajs, spec.all_t_indices, "val")["motion_pred"]
    assert out["pair_names"] == whole["pair_names"]
    assert torch.equal(out["motion_pred"], whole["motion_pred"]) and torch.equal(out["reconst_pred"], whole["reconst_pred"])
    for k, r in whole["rollout_trajs"].items():
        assert torch.equal(out["rollout_trajs"][k]["traj"], r["traj"]) and torch.equal(out["rollout_trajs"][k]["vel"], r["vel"])
    again = model.decode_batch(model.encode_scene(batch), model.encode_prompt(batch), batch, "val")["motion_pred"]
    assert torch.equal(again["motion_pred"], whole["motion_pred"])
    with pytest.raises(ValueError, match="all_t_indices"):
        model.rollout_batch(batch, scene_embs, policy_emds, ids, model.init_agent_trajs(ids, batch), [0, 10], "val")


def test_fut_obs_frames_are_matched_by_agent_id(model):
    """Every fut_obs frame lists its own agents (get_center_obs drops agents that have left, format_utils.py:383-388):
    a frame that omits a log-replay agent and lists the others in another order must equal the same frame with
    that agent masked in place; a policy agent missing from a frame is an error."""
    name = "small_replay_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    scene = synth.make_scene(spec, **kw)
    pm = scene["prompt_mask"].astype(bool)
    seen = scene["obs_mask"].all(-1).any(-1)
    b0 = 0
    replay = [n for n in range(pm.shape[1]) if seen[b0, n] and not pm[b0, n]]
    assert len(replay) >= 2
    leaver = replay[0]
    masked = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in scene.items()}
    masked["fut_obs_mask"][2:, b0, leaver] = False                      # gone from the log from the 3rd later replan on
    masked["fut_obs_input"][2:, b0, leaver] = np.nan
    want = model(make_batch(masked, spec), "val")["motion_pred"]
    batch = make_batch(scene, spec)
    for k, t in enumerate(sorted(batch.extras["fut_obs"].keys())):
        if k < 2:
            continue
        fr = batch.extras["fut_obs"][t]
        n_b = len(fr["agent_ids"][b0])
        order = [n for n in reversed(range(n_b)) if n != leaver]        # drop the leaver, reverse the others
        for key in ("input", "mask", "position", "heading"):
            rows = fr[key][b0, order].clone()
            fr[key][b0] = 0 if key != "mask" else False
            fr[key][b0, :len(order)] = rows
        fr["agent_ids"] = [list(a) for a in fr["agent_ids"]]
        fr["agent_ids"][b0] = [fr["agent_ids"][b0][n] for n in order]
    got = model(batch, "val")["motion_pred"]
    assert got["pair_names"] == want["pair_names"] and torch.equal(got["motion_pred"], want["motion_pred"])
    t_last = sorted(batch.extras["fut_obs"].keys())[-1]
    lost = make_batch(scene, spec)
    pol = int(np.nonzero(pm[b0])[0][0])
    fr = lost.extras["fut_obs"][t_last]
    fr["agent_ids"] = [list(a) for a in fr["agent_ids"]]
    fr["agent_ids"][b0] = [a for n, a in enumerate(fr["agent_ids"][b0]) if n != pol]
    for key in ("input", "mask", "position", "heading"):
        keep = [n for n in range(len(fr["agent_ids"][b0]) + 1) if n != pol]
        rows = fr[key][b0, keep].clone()
        fr[key][b0, :len(keep)] = rows
    with pytest.raises(ValueError, match="missing from fut_obs"):
        model(lost, "val")


def _listed_form(batch):
    """The batch as the reference's dataset code builds it: init_obs and every fut_obs frame list ONLY the agents
    that are in the scene at that step (get_center_obs, format_utils.py:383-388), compacted, with their ids."""
    ex = batch.extras

    def compact(fr):
        m = fr["mask"].all(-1).any(-1)                                       # [B, N] in the scene at this step
        B = m.shape[0]
        keep = [torch.nonzero(m[b])[:, 0].tolist() for b in range(B)]
        Nn = max(len(k) for k in keep)
        out = dict(fr)
        for key in ("input", "mask", "position", "heading"):
            t = fr[key]
            new = torch.zeros((B, Nn) + tuple(t.shape[2:]), dtype=t.dtype)
            if key == "input":
                new = new * float("nan")
            for b in range(B):
                new[b, :len(keep[b])] = t[b, keep[b]]
            out[key] = new
        out["agent_ids"] = [[fr["agent_ids"][b][n] for n in keep[b]] for b in range(B)]
        return out

    ex["init_obs"] = compact(ex["init_obs"])
    for t in list(ex["fut_obs"].keys()):
        ex["fut_obs"][t] = compact(ex["fut_obs"][t])
    return batch


def test_agents_entering_the_scene_vs_fixture(model):
    """Agents that are not in the scene at the initial step and enter with a later fut_obs frame: the reference-made
    fixture (every agent listed everywhere, presence by mask) and the dataset's own form of the same batch (each
    frame lists only the agents present, new ids appear later) give the same rollout."""
    name = "small_enter_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    seen0 = scene["obs_mask"].all(-1).any(-1)
    ever = seen0 | scene["fut_obs_mask"].all(-1).any(-1).any(0)
    assert (ever & ~seen0).sum() >= 4                                           # the case really has entering agents
    masked_form = model(make_batch(scene, spec), "val")["motion_pred"]
    A = model.engine.num_policy_agents
    assert model.engine.num_agents == int(ever.sum()) and int(model.engine.live0_rows.sum()) == int(seen0.sum())
    assert err(masked_form["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < 1e-4
    floor = g["fp32_floor"][0]
    pm = scene["prompt_mask"].astype(bool)
    for b in range(2):
        for n in np.nonzero(pm[b])[0]:
            assert err(masked_form["rollout_trajs"][f"{b}-a{n}"]["traj"].numpy(), g["traj"][b, n]) < 3 * floor + 1e-4
    listed = _listed_form(make_batch(scene, spec))
    assert len(listed.extras["init_obs"]["agent_ids"][0]) == int(seen0[0].sum())
    listed_form = model(listed, "val")["motion_pred"]
    assert listed_form["pair_names"] == masked_form["pair_names"]
    # the entering agents now sit in other rows (new slots behind the initial ones): same sets, another summation order
    assert err(listed_form["motion_pred"][:2 * A].numpy(), masked_form["motion_pred"][:2 * A].numpy()) < 1e-4
    for k, r in masked_form["rollout_trajs"].items():
        assert err(listed_form["rollout_trajs"][k]["traj"].numpy(), r["traj"].numpy()) < 3 * floor + 1e-4


@pytest.mark.parametrize("variant", ["fixed_pe", "learnable_pe", "mlp_head", "cluster_head"])
def test_staged_components_and_stateless_policy(model, variant):
    """scene_encoder(...) -> decoder(...) -> policy(...) with the reference's argument layouts (with the fixed Fourier
    rel-PE and with the learnable one: the stateless policy call builds its own edge sets and rows)."""
    spec = SMALL_SPEC
    if variant != "fixed_pe":
        from prosim_amd import modules
        spec = {"learnable_pe": SMALL_SPEC.replace(enc_learnable_pe=True, dec_learnable_pe=True, pol_learnable_pe=True),
                "mlp_head": SMALL_SPEC.replace(k_pred_mode="mlp", motion_k=2),          # TRAJ.PRED_MODE through ps_policy_forward
                "cluster_head": SMALL_SPEC.replace(k_pred_mode="cluster", motion_k=3)}[variant]
        model = modules.registry.get_model("prosim_policy_relpe_T_step_temporal_close_loop")(spec, weights.init_weights(spec, 0)).eval()
    scene = synth.make_scene(spec, 12, 40, batch=2, seed=4, ragged=True)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
    batch = make_batch(scene, spec).extras
    se = model.scene_encoder(batch["init_obs"], batch["init_map"])
    for k in ("obs_mask", "map_mask", "scene_batch_idx", "scene_type", "scene_pos", "scene_ori", "scene_tokens", "max_map_num", "max_agent_num"):
        assert k in se                                                           # attn_fusion.py:121-134
    assert err(se["scene_tokens"].numpy(), o["trace"]["scene_tokens"].numpy()) < 1e-4
    assert torch.equal(se["scene_type"], (torch.arange(se["scene_type"].numel()) >= se["_n_map_tokens"]).long())
    pe = model.decoder(se, batch["prompt"]["motion_pred"])
    pm = torch.from_numpy(scene["prompt_mask"].astype(bool))
    dm = batch["prompt"]["motion_pred"]["prompt_mask"]                            # dense prompt rows (collate layout)
    assert pe["emd"].shape == (2, dm.shape[1], 128)
    assert err(pe["emd"][dm].numpy(), o["policy_emd"][pm].numpy()) < 1e-4
    # policy.forward on reference-style padded tokens (traj_sam.py:356-400 layout), replan 0
    B, N = pm.shape
    Mv = se["_n_map_tokens"]
    def padded(tok, pos, ori, bidx, S):
        inp, mask = torch.zeros(B, S, 128), torch.zeros(B, S, dtype=torch.bool)
        p, o_ = torch.zeros(B, S, 2), torch.zeros(B, S, 1)
        for b in range(B):
            sel = bidx == b
            n = int(sel.sum())
            inp[b, :n], p[b, :n], o_[b, :n], mask[b, :n] = tok[sel], pos[sel], ori[sel], True
        return dict(input=inp, mask=mask, pos=p, ori=o_)
    tok = se["scene_tokens"]
    bo = padded(tok[Mv:], se["scene_pos"][Mv:], se["scene_ori"][Mv:], se["scene_batch_idx"][Mv:], N)
    bm = padded(tok[:Mv], se["scene_pos"][:Mv], se["scene_ori"][:Mv], se["scene_batch_idx"][:Mv], scene["map_mask"].shape[1])
    bidx = orc._flat_batch_idx(pm)
    pol_emd = dict(emd=pe["emd"][dm], agent_type=torch.from_numpy(scene["agent_type"])[pm], batch_idx=bidx)
    pos = dict(position=torch.from_numpy(scene["obs_pos"])[pm], heading=orc.wrap_angle(torch.from_numpy(scene["obs_head"])[pm])[:, None])
    names = [f"{int(b)}-x{i}-0" for i, b in enumerate(bidx)]
    out = model.policy(pol_emd, bo, bm, pos, names, None)
    A = int(pm.sum())
    assert out["latent_state"] is None and model.policy.format_latent_state({}, [names]) is None
    assert err(out["motion_pred"].numpy(), o["motion_pred"][:A].numpy()) < 1e-4
    assert out["motion_pred"].shape == (A, spec.motion_k, spec.target_steps, spec.state_dim)
    assert torch.equal(out["motion_prob"], torch.ones(A, spec.motion_k))
    with pytest.raises(AssertionError):
        model.policy(pol_emd, bo, bm, pos, names[:-1], None)
    # update_scene_emb (attn_fusion.py:238-252, FUSION 'replace'): agents re-encoded from a new observation, map tokens kept
    rng = np.random.RandomState(3)
    new_obs = dict(batch["init_obs"])
    new_obs["input"] = batch["init_obs"]["input"] + torch.from_numpy(rng.uniform(-0.3, 0.3, tuple(batch["init_obs"]["input"].shape)).astype(np.float32))
    new_obs["position"] = batch["init_obs"]["position"] + 1.5
    new_obs["heading"] = batch["init_obs"]["heading"] + 0.2
    se2 = model.scene_encoder.update_scene_emb(se, new_obs, batch["init_obs"]["agent_ids"])
    Wt = orc.W(w)
    emb, valid = orc.encode_obs(Wt, spec, torch.nan_to_num(new_obs["input"]), new_obs["mask"])
    assert torch.equal(se2["scene_tokens"][:Mv], se["scene_tokens"][:Mv])                 # map tokens reused
    assert err(se2["scene_tokens"][Mv:].numpy(), emb[valid].numpy()) < 1e-4
    assert err(se2["scene_pos"][Mv:].numpy(), new_obs["position"][valid].numpy()) == 0
    assert err(se2["scene_tokens"][Mv:].numpy(), se["scene_tokens"][Mv:].numpy()) > 1e-2   # and they did change
    # ... with ANOTHER agent set (_replace_old_obs :205-236 takes whatever the new observation lists): per scene the first
    # agent has left, the others arrive in reversed order, and one that was never seen enters (a copy of an agent, moved)
    ids0 = batch["init_obs"]["agent_ids"]
    B_, N_ = new_obs["input"].shape[:2]
    cnt = new_obs["mask"].all(-1).any(-1).sum(1)
    ch = {k: torch.zeros_like(new_obs[k]) for k in ("input", "mask", "position", "heading")}
    ch_ids = []
    for b_ in range(B_):
        order = list(range(int(cnt[b_]) - 1, 0, -1))                     # agents 1.. reversed; agent 0 is gone
        n = len(order)
        for k in ch:
            ch[k][b_, :n] = new_obs[k][b_, order]
            ch[k][b_, n] = new_obs[k][b_, order[0]]                          # the newcomer: same history ...
        ch["position"][b_, n] += 7.0                                         # ... elsewhere
        ch_ids.append([ids0[b_][j] for j in order] + ["newcomer"])
    changed = dict(new_obs, agent_ids=ch_ids, **ch)
    se3 = model.scene_encoder.update_scene_emb(se2, changed, ids0)
    emb3, valid3 = orc.encode_obs(Wt, spec, torch.nan_to_num(changed["input"]), changed["mask"])
    assert int(valid3.sum()) == int(cnt.sum()) and se3["scene_tokens"].shape[0] == Mv + int(valid3.sum())
    assert torch.equal(se3["scene_tokens"][:Mv], se["scene_tokens"][:Mv])               # the map tokens are still the encoder's
    assert err(se3["scene_tokens"][Mv:].numpy(), emb3[valid3].numpy()) < 1e-4
    assert err(se3["scene_pos"][Mv:].numpy(), changed["position"][valid3].numpy()) == 0
    assert torch.equal(se3["scene_batch_idx"][Mv:], orc._flat_batch_idx(valid3))
    if variant != "fixed_pe":
        model.engine.close()
