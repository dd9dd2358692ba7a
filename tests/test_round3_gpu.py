"""Round-3 GPU tests: the per-step model methods (step_env / decode_output / step_agent_traj / get_action), action noise and the
PRED_GMM head through the registry-level model, the engine on a batch made by the REFERENCE's own formatters, the stateless
policy call next to a captured rollout (the exchange buffers no longer exist), replicas that differ through noise."""
import os

import numpy as np
import pytest
import torch

from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from oracle import prosim_oracle as orc
from oracle import ref_batch as rh
from golden_cases import FULL_CASES, SPECS, GOLD

pytestmark = pytest.mark.gpu
TOL = 1e-4


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_per_step_methods_equal_forward_and_the_fixture():
    """ProSim's call sequence by hand (traj_sam.py:59-71, :144-176): encode_scene, encode_prompt, generate_policy,
    init_agent_trajs, then per replan step_env -> decode_output -> step_agent_traj, then _process_rollout -- bit-equal to
    forward(batch) (same kernels, same order) and, replan by replan, the reference fixture's motion_pred."""
    from prosim_amd import modules
    name = "small_replay_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    model = modules.ProSimHip(spec, weights.init_weights(spec, wseed))
    try:
        batch = rh.make_batch(scene, spec)
        whole = model.forward(batch, "val")["motion_pred"]
        emb = model.encode_scene(batch)
        pol = model.generate_policy(batch, emb, model.encode_prompt(batch))
        ids = {"motion_pred": batch.extras["prompt"]["motion_pred"]["agent_ids"]}
        tr = model.init_agent_trajs(ids, batch)
        A = int(scene["prompt_mask"].sum())
        assert tr["motion_pred"]["traj"].shape[2] == spec.hist_steps and tr["motion_pred"]["last_step"] == spec.hist_steps
        all_t = list(spec.all_t_indices)
        floor = g["fp32_floor"][2]
        for i, t in enumerate(all_t):
            emb, a_pos = model.step_env(emb, tr, batch, ids, t, all_t)
            assert a_pos["position"].shape[-1] == 2 and a_pos["heading"].shape[-1] == 1
            if i == 0:   # the rollout starts from the observed poses
                p0 = np.concatenate([scene["obs_pos"][b][scene["prompt_mask"][b]] for b in range(scene["obs_pos"].shape[0])])
                got = np.concatenate([a_pos["position"][b, :int(scene["prompt_mask"][b].sum())].numpy() for b in range(scene["obs_pos"].shape[0])])
                assert err(got, p0) < 1e-6
            out = model.decode_output(pol, emb, ids, batch, a_pos, t, None)["motion_pred"]
            assert out["motion_pred"].shape == (A, 1, spec.target_steps, spec.state_dim) and len(out["pair_names"]) == A
            assert out["pair_names"][0].endswith(f"-{t}")
            assert err(out["motion_pred"].numpy(), g["motion_pred"][i * A:(i + 1) * A]) < 3 * floor + TOL
            assert np.array_equal(out["motion_pred"].numpy(), whole["motion_pred"][i * A:(i + 1) * A].numpy())
            tr = model.step_agent_traj(tr, {"motion_pred": out}, ids, t, "val")
            assert tr["motion_pred"]["last_step"] == spec.hist_steps + (i + 1) * spec.replan_freq
        with pytest.raises(RuntimeError):
            model.step_env(emb, tr, batch, ids, 0, all_t)            # replans cannot be re-run out of order
        staged = model._process_rollout(batch.extras, model._shared.scene)["motion_pred"]
        for k_, r in whole["rollout_trajs"].items():
            assert np.array_equal(r["traj"].numpy(), staged["rollout_trajs"][k_]["traj"].numpy())
        # the mirrored a_traj holds the same steps (policy-agent order)
        b0 = [n for n in np.nonzero(scene["prompt_mask"][0])[0]]
        assert np.array_equal(tr["motion_pred"]["traj"][0, 0, spec.hist_steps:].numpy(), whole["rollout_trajs"][f"0-a{b0[0]}"]["traj"].numpy())
    finally:
        model.close()


def test_step_agent_traj_takes_an_edited_prediction():
    """A caller may change the policy's output between decode_output and step_agent_traj (the reference's update runs on
    whatever it is handed, traj_sam.py:311-347): a prediction scaled to half its steps moves the agents half as far, and the
    next replan starts from that state (ps_set_state)."""
    from prosim_amd import modules
    spec = SMALL_SPEC
    scene = synth.make_scene(spec, 12, 96, batch=2, seed=21, goal=True, ragged=True)
    model = modules.ProSimHip(spec, weights.init_weights(spec, 0))
    try:
        batch = rh.make_batch(scene, spec)
        emb = model.encode_scene(batch)
        pol = model.generate_policy(batch, emb, model.encode_prompt(batch))
        ids = {"motion_pred": batch.extras["prompt"]["motion_pred"]["agent_ids"]}
        tr = model.init_agent_trajs(ids, batch)
        all_t = list(spec.all_t_indices)
        emb, a_pos = model.step_env(emb, tr, batch, ids, 0, all_t)
        out = model.decode_output(pol, emb, ids, batch, a_pos, 0, None)["motion_pred"]
        edited = dict(out, motion_pred=out["motion_pred"] * torch.tensor([0.5, 0.5, 1.0, 1.0, 1.0]))
        tr = model.step_agent_traj(tr, {"motion_pred": edited}, ids, 0, "val")
        H, S = spec.hist_steps, spec.replan_freq
        # first step of agent 0: rotate by the last history heading (0 in its own frame) -> exactly half the predicted step
        assert err(tr["motion_pred"]["traj"][0, 0, H, :2].numpy(), 0.5 * out["motion_pred"][0, 0, 0, :2].numpy()) < 1e-6
        dev = model.engine.padded("traj")
        n0 = int(np.nonzero(scene["prompt_mask"][0])[0][0])
        assert err(dev[0, n0, :S], tr["motion_pred"]["traj"][0, 0, H:H + S].numpy()) < 1e-6      # the device holds the edited state
        emb, a_pos = model.step_env(emb, tr, batch, ids, all_t[1], all_t)
        assert err(a_pos["position"][0, 0].numpy(), scene["obs_pos"][0, n0] + tr["motion_pred"]["traj"][0, 0, H + S - 1, :2].numpy()) < 1e-5
        out1 = model.decode_output(pol, emb, ids, batch, a_pos, all_t[1], None)["motion_pred"]
        assert torch.isfinite(out1["motion_pred"]).all()
    finally:
        model.close()


def test_action_noise_and_gmm_head_replay_the_reference_stream():
    """RANDOM_NOISE_STD > 0 + PRED_GMM through the registry-level model: ProSimHip draws the noise with the reference's own
    torch.randn_like call (and consumes step_agent_traj's torch.randint), so the fixture's seed reproduces the fixture's
    draws, and the rollout -- velocity from columns 6:8 of the 8-wide state -- its trajectories."""
    from prosim_amd import modules
    name = "small_noise_gmm_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    assert spec.state_dim == 8 and spec.vel_col == 6
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    model = modules.ProSimHip(spec, weights.init_weights(spec, wseed))
    try:
        batch = rh.make_batch(scene, spec)
        torch.manual_seed(int(g["torch_seed"]))
        out = model.forward(batch, "val")["motion_pred"]
        pm = scene["prompt_mask"].astype(bool)
        assert np.array_equal(model._last_action_noise[:, pm], g["action_noise"][:, pm])
        assert float(np.abs(g["action_noise"][:, pm]).max()) > 0.05
        A = int(pm.sum())
        assert out["motion_pred"].shape == (spec.n_replans * A, 1, 10, 8)
        assert err(out["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < TOL
        floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
        for b in range(pm.shape[0]):
            for n in np.nonzero(pm[b])[0]:
                r = out["rollout_trajs"][f"{b}-a{n}"]
                assert err(r["traj"].numpy(), g["traj"][b, n]) < 3 * floor["traj"] + TOL
                assert err(r["vel"].numpy(), g["vel"][b, n]) < 3 * floor["vel"] + TOL
        # without the noise the same model lands elsewhere (the table is really applied) ...
        quiet = modules.ProSimHip(spec.replace(action_noise_std=0.0), weights.init_weights(spec, wseed))
        try:
            q = quiet.forward(batch, "val")["motion_pred"]
            assert err(q["motion_pred"][:A].numpy(), g["motion_pred"][:A]) > 1e-2
        finally:
            quiet.close()
    finally:
        model.close()


def test_replicas_differ_through_action_noise():
    """parallel_rollout_batch with TOP_K = K = 1: the replicas of a scene are identical rollouts unless RANDOM_NOISE_STD makes
    them differ (act_decoder.py:113-115) -- every replica gets its own rows of the noise table; against the oracle on the
    replicated batch with the same table."""
    from prosim_amd.engine import Engine
    from prosim_amd.postprocess import replicate_scene
    spec = SMALL_SPEC.replace(action_noise_std=0.1)
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 10, 96, batch=1, seed=23, goal=True)
    M, N = 4, scene["prompt_mask"].shape[1]
    rng = np.random.default_rng(5)
    noise = (rng.standard_normal((spec.n_replans, M, N, 1, spec.target_steps, 2)) * spec.action_noise_std).astype(np.float32)
    eng = Engine(spec, w)
    try:
        eng.set_replicas(M)
        eng.set_scene(scene)
        eng.set_action_noise(noise)
        eng.rollout()
        traj = eng.padded("traj")                                             # [M, N, 80, 4]
        assert traj.shape[0] == M
        d01 = np.abs(traj[0] - traj[1]).max()
        assert d01 > 0.05                                                     # the replicas really differ
        tiled = replicate_scene(scene, M)
        tiled["action_noise"] = noise
        with torch.no_grad():
            o64 = orc.rollout(w, spec, tiled, dtype=torch.float64)
            o32 = orc.rollout(w, spec, tiled, dtype=torch.float32)
        floor = float((o32["traj"].double() - o64["traj"]).abs().max())
        assert err(eng.get("motion_pred")[0], o64["motion_pred"][:eng.num_agents].numpy()) < TOL
        assert err(traj, o64["traj"].numpy()) < 3 * floor + TOL
    finally:
        eng.close()


@pytest.mark.parametrize("scene_name", ["scene_1"])
def test_engine_on_the_reference_formatted_batch(scene_name):
    """SURVEY section 8 row (f3), device side: the batch the REFERENCE's own formatters made of the demo cache
    (tests/golden/ref_format_scene_1.npz: get_center_obs, get_center_vec_init_map, prompt_for_batch) goes into the engine
    as it is -- 37 agents, 504 lane chunks -- against the fp64 oracle on the same arrays."""
    from prosim_amd.engine import Engine
    g = np.load(os.path.join(GOLD, f"ref_format_{scene_name}.npz"))
    spec = DEMO_SPEC.replace(max_steps=20)
    used = g["map_mask"][0].any(-1)
    Mu = int(np.nonzero(used)[0][-1]) + 1
    ids = list(g["obs_ids"])
    N = len(ids)
    sel = [ids.index(a) for a in g["prompt_ids"]]
    prompt = np.zeros((1, N, spec.prompt_dim), np.float32)
    pmask = np.zeros((1, N), bool)
    types = np.ones((1, N), np.int64)
    prompt[0, sel], pmask[0, sel], types[0, sel] = g["prompt"][0], True, g["prompt_type"][0]
    scene = dict(map_input=np.where(g["map_mask"][..., None], g["map_input"], 0.0)[:, :Mu].astype(np.float32), map_mask=g["map_mask"][:, :Mu],
                 map_pos=g["map_pos"].reshape(1, -1, 2)[:, :Mu], map_head=g["map_head"].reshape(1, -1)[:, :Mu],
                 obs_input=g["obs_input"], obs_mask=g["obs_mask"], obs_pos=g["obs_pos"], obs_head=g["obs_head"],
                 prompt=prompt, prompt_mask=pmask, agent_type=types)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
        o32 = orc.rollout(w, spec, scene, dtype=torch.float32, collect=True)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.encode_scene()
        tok = eng.get("scene_tokens")
        e64 = np.abs(tok - o64["trace"]["scene_tokens"].numpy()).max(1)
        e32 = np.abs(tok - o32["trace"]["scene_tokens"].numpy()).max(1)
        assert np.minimum(e32, e64).max() < TOL and (e64 < TOL).mean() > 0.8   # (antiparallel lane chunks: the +-pi cut, see test_hip_parity)
        eng.rollout()
        A = eng.num_agents
        assert A == N
        assert err(eng.get("motion_pred")[0], o64["motion_pred"][:A].numpy()) < TOL
        d = np.abs(eng.padded("traj") - o64["traj"].numpy())[0].reshape(N, -1).max(1)
        floor = float((o32["traj"].double() - o64["traj"]).abs().max())
        assert d.max() < 3 * floor + TOL and np.median(d) < TOL, (d, floor)
    finally:
        eng.close()


def test_stateless_policy_call_beside_a_captured_rollout():
    """ps_policy_forward with MORE rows than the uploaded scene (throughput mode: k_chain16) between two replays of the
    scene's captured graph: the fused chains keep their phase exchange in LDS, so the call cannot move a buffer under the
    graph; the replay returns the same bits as before."""
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 12, 96, batch=2, seed=31, goal=True, ragged=True)
    big = synth.make_scene(spec, 48, 160, batch=3, seed=32)
    eng = Engine(spec, w)
    try:
        eng.set_chain_rows(16)
        eng.set_scene(scene)
        eng.rollout(); eng.sync()
        eng.rollout(); eng.sync()                                              # (replayed from the graph)
        before = eng.padded("traj").copy()
        # explicit tokens of another, larger batch
        rng = np.random.default_rng(3)
        Na, Nm, A = 144, 480, 144
        mp, fused = eng.policy_forward(3, rng.standard_normal((Na, 128)).astype(np.float32), rng.uniform(-50, 50, (Na, 2)).astype(np.float32),
                                       rng.uniform(-3, 3, Na).astype(np.float32), np.repeat(np.arange(3), 48).astype(np.int32),
                                       rng.standard_normal((Nm, 128)).astype(np.float32), rng.uniform(-50, 50, (Nm, 2)).astype(np.float32),
                                       rng.uniform(-3, 3, Nm).astype(np.float32), np.repeat(np.arange(3), 160).astype(np.int32),
                                       rng.standard_normal((A, 128)).astype(np.float32), rng.uniform(-50, 50, (A, 2)).astype(np.float32),
                                       rng.uniform(-3, 3, A).astype(np.float32), np.ones(A, np.int32), np.repeat(np.arange(3), 48).astype(np.int32))
        assert np.isfinite(mp).all() and np.isfinite(fused).all()
        eng.rollout(); eng.sync()
        assert np.array_equal(eng.padded("traj"), before)
    finally:
        eng.close()


@pytest.mark.parametrize("rows", [0, 16])
def test_encoder_s2s_layers_on_k_chain16(rows):
    """ps_set_chain_impl(3): the scene encoder's s2s layers as one-step k_chain16 launches (k | v projection + chain) instead of
    the split k_node / k_edge_small path -- the same layer math, another fp32 evaluation order.  Open-loop quantities against the
    fp64 oracle at 1e-4 (scene tokens, generator output, replan 0); the closed loop per agent with the usual branch-cut allowance
    (this is NOT the default path: on the benchmark workload it flips six agents of one scene)."""
    from prosim_amd.engine import Engine
    from oracle_cache import oracle64
    spec = DEMO_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.baseline_scene(spec, 3, seed=1, batch=2)
    o64 = oracle64("baseline_cfg3_b2_s1", spec, w, scene, collect=True)
    eng = Engine(spec, w)
    try:
        eng.set_chain_impl(3)
        eng.set_chain_rows(rows)
        eng.set_scene(scene)
        eng.encode_scene()
        assert err(eng.get("scene_tokens"), o64["trace"]["scene_tokens"].numpy()) < TOL
        eng.generate_policy()
        pm = scene["prompt_mask"].astype(bool)
        assert err(eng.get("policy_emd"), o64["policy_emd"][torch.from_numpy(pm)].numpy()) < 2 * TOL
        eng.rollout()
        A = eng.num_agents
        assert err(eng.get("motion_pred")[0], o64["motion_pred"][:A].numpy()) < TOL
        d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
        assert (d < TOL).mean() >= 0.97 and np.median(d) < 3e-5, ((d < TOL).mean(), np.median(d), d.max())
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["small_cluster_b2", "small_mlphead_b2"])
def test_pred_mode_cluster_and_mlp_through_the_model(name):
    """TRAJ.PRED_MODE 'cluster' (anchors folded from the goal-cluster file) and 'mlp' (all K modes from motion_head, no
    CG_decode) through the registry-level model with TOP_K = K: the fixture's seed reproduces the reference's mode draws, and
    the engine its trajectories; motion_pred carries K modes per agent."""
    from prosim_amd import modules
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    model = modules.ProSimHip(spec, weights.init_weights(spec, wseed))
    try:
        batch = rh.make_batch(scene, spec)
        torch.manual_seed(int(g["torch_seed"]))
        out = model.forward(batch, "val")["motion_pred"]
        pm = scene["prompt_mask"].astype(bool)
        assert np.array_equal(model._last_mode_choice[:, pm], g["mode_choice"][:, pm])
        assert len(np.unique(g["mode_choice"][:, pm])) == spec.motion_k           # every mode is followed somewhere
        A = int(pm.sum())
        assert out["motion_pred"].shape == (spec.n_replans * A, spec.motion_k, spec.target_steps, spec.state_dim)
        assert err(out["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < TOL
        floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
        for b in range(pm.shape[0]):
            for n in np.nonzero(pm[b])[0]:
                r = out["rollout_trajs"][f"{b}-a{n}"]
                assert err(r["traj"].numpy(), g["traj"][b, n]) < 3 * floor["traj"] + TOL
                assert err(r["vel"].numpy(), g["vel"][b, n]) < 3 * floor["vel"] + TOL
    finally:
        model.close()


def test_without_pred_vel_the_rollout_keeps_no_velocity_track():
    """TRAJ.PRED_VEL False through the registry-level model: 3-wide states, rollout_trajs without 'vel' (traj_sam.py:592-593), the
    observation's velocity / acceleration columns from position differences -- the fixture's trajectories."""
    from prosim_amd import modules
    name = "small_novel_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    assert spec.state_dim == 3 and spec.vel_col == -1
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    model = modules.ProSimHip(spec, weights.init_weights(spec, wseed))
    try:
        out = model.forward(rh.make_batch(scene, spec), "val")["motion_pred"]
        pm = scene["prompt_mask"].astype(bool)
        A = int(pm.sum())
        assert out["motion_pred"].shape == (spec.n_replans * A, 1, spec.target_steps, 3)
        assert err(out["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < TOL
        floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
        for b in range(pm.shape[0]):
            for n in np.nonzero(pm[b])[0]:
                r = out["rollout_trajs"][f"{b}-a{n}"]
                assert "vel" not in r
                assert err(r["traj"].numpy(), g["traj"][b, n]) < 3 * floor["traj"] + TOL
        assert float(np.abs(model.engine.padded("vel")).max()) == 0.0
    finally:
        model.close()


def test_without_goal_pred_loss_there_is_no_reconst_pred():
    """LOSS.ROLLOUT_TRAJ.USE_GOAL_PRED_LOSS False through the registry-level model: a checkpoint without pred_mlp loads, the output
    has no 'reconst_pred' (act_decoder.py:128-130), the per-replan calls neither; trajectories as the fixture's."""
    from prosim_amd import modules
    name = "small_nogoalloss_b2"
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    w = weights.init_weights(spec, wseed)
    assert not any(".pred_mlp." in k for k in w)
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    scene = synth.make_scene(spec, **kw)
    model = modules.ProSimHip(spec, w)
    try:
        out = model.forward(rh.make_batch(scene, spec), "val")["motion_pred"]
        assert "reconst_pred" not in out and {"motion_pred", "motion_prob", "pair_names", "rollout_trajs"} <= set(out)
        pm = scene["prompt_mask"].astype(bool)
        A = int(pm.sum())
        assert err(out["motion_pred"][:A].numpy(), g["motion_pred"][:A]) < TOL
        floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
        for b in range(pm.shape[0]):
            for n in np.nonzero(pm[b])[0]:
                assert err(out["rollout_trajs"][f"{b}-a{n}"]["traj"].numpy(), g["traj"][b, n]) < 3 * floor["traj"] + TOL
    finally:
        model.close()


@pytest.mark.parametrize("rows", [0, 16])
def test_the_timed_batch_against_the_reference_itself(rows):
    """bench.py's batch (8 x configs[2], seeds 0..7) in latency mode and in throughput mode (16 rows per workgroup: the headline's
    kernels) against the REFERENCE's own fp32 forward of every scene (the model never mixes batch elements, so scene b of the batch is
    fixture demo_cfg2 seed b): every one of the 1024 agents within the fixture test's bar, and how many within 1e-4 outright."""
    from prosim_amd.engine import Engine
    from parity_table import record, per_agent
    spec = DEMO_SPEC
    w = weights.init_weights(spec, 0)
    parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
    scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
                 {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
    eng = Engine(spec, w)
    try:
        eng.set_chain_rows(rows)
        eng.set_scene(scene)
        eng.rollout()
        traj = eng.padded("traj")
        d_all = []
        for b in range(8):
            g = np.load(os.path.join(GOLD, "ref_standins_demo_cfg2_b1.npz" if b == 0 else f"ref_standins_demo_cfg2_seed{b}.npz"))
            d = np.abs(traj[b] - g["traj"][0]).max(axis=(1, 2))                 # per agent, over the 80 steps
            assert d.max() < 3 * float(g["fp32_floor"][0]) + TOL, (b, float(d.max()))
            d_all.append(d)
        d_all = np.concatenate(d_all)
        record(f"bench_workload_vs_reference/rows{rows}", **per_agent(d_all))
        print(f"batch vs the reference, rows {rows}: max {d_all.max():.2e} median {np.median(d_all):.2e} within 1e-4: {(d_all < 1e-4).sum()} / {d_all.size}")
        assert (d_all < 1e-4).mean() >= 0.995 and np.median(d_all) < 3e-5
    finally:
        eng.close()
