"""GPU parity tests (pytest -m gpu): the HIP engine, called through the C ABI
(include/prosim_hip.h via prosim_amd/engine.py), against the oracle and the golden fixtures.

Tolerances.  north_star asks for 1e-4 on identical seeds.  fp32 itself cannot hold 1e-4 over a
closed loop on every scene: rounding noise is amplified ~1.5-2x per replan (DESIGN.md "fp32 noise
floor"; the reference's own fp32 run sits 6e-4 from the fp64 restatement on the worst fixture).
So: (a) every OPEN-loop quantity (encoders, generator, one policy step from a given state) is held
to 1e-4 absolute; (b) closed-loop trajectories at the BASELINE sizes are held to 1e-4 against the
fp64 restatement; (c) the small chaotic fixtures are held to 3x the measured fp32 floor + 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from oracle import prosim_oracle as orc
from golden_cases import FULL_CASES, REPORT_ONLY, SPECS, digest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.fixture(scope="module")
def demo_engine():
    from prosim_amd.engine import Engine
    eng = Engine(DEMO_SPEC, weights.init_weights(DEMO_SPEC, 0))
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def small_engine():
    from prosim_amd.engine import Engine
    eng = Engine(SMALL_SPEC, weights.init_weights(SMALL_SPEC, 0))
    yield eng
    eng.close()


def test_library_is_the_hip_build():
    from prosim_amd import engine
    lib = engine.load_library()
    assert os.path.basename(engine.lib_path()) == "libprosim_hip.so"
    for sym in engine.EXPORTS:
        assert hasattr(lib, sym)


# ------------------------------------------------------------------ primitives vs reference-pure fixtures
def test_pointnet_fourier_wrap_ref_pure(demo_engine):
    g = np.load(os.path.join(GOLD, "ref_pure_primitives.npz"))
    for which, tag in ((0, "map"), (1, "obs")):
        x, m = g[f"pointnet_{tag}_x"], g[f"pointnet_{tag}_mask"]
        y = demo_engine.test_pointnet(which, x.reshape(-1, *x.shape[2:]), m.reshape(-1, m.shape[2]))
        valid = m.reshape(-1, m.shape[2]).any(-1)
        assert err(y[valid], g[f"pointnet_{tag}_y"].reshape(-1, 128)[valid]) < 1e-5
    assert err(demo_engine.test_fourier(g["fourier_x"]), g["fourier_y"]) < 1e-6
    assert np.array_equal(demo_engine.test_wrap(g["wrap_x"]), g["wrap_y"])   # bit-exact (fmod path)
    edge = np.array([0.0, np.pi, -np.pi, 3 * np.pi, -3 * np.pi, 1e-7, -1e-7, 6.2831855, 100.0, -100.0], np.float32)
    assert np.array_equal(demo_engine.test_wrap(edge), orc.wrap_angle(torch.from_numpy(edge)).numpy())


def _rnorm(r):
    return ((r - r.mean(-1, keepdim=True)) / torch.sqrt(r.var(-1, unbiased=False, keepdim=True) + 1e-5)).numpy()


@pytest.mark.parametrize("group,prefix,bip", [("s2s", "scene_encoder.s2s_attn_layers.1", False),
                                              ("a2p", "policy.act_decoder.a2p_attn_layers.0", True),
                                              ("cond", "condition_transformers.policy_decoder.condition_attn.attn_layers.0", False)])
@pytest.mark.parametrize("shape", [(20, 7, 60), (300, 37, 2000), (50, 9, 0), (900, 5, 3500)])
def test_attention_layer_vs_oracle(small_engine, group, prefix, bip, shape):
    """One AttentionLayer on a random graph (incl. destinations with no edge, degree > 700)."""
    Ns, Nd, E = shape
    g = torch.Generator().manual_seed(Ns + Nd + E)
    Wt = orc.W(weights.init_weights(SMALL_SPEC, 0))
    xs, xd = torch.randn(Ns, 128, generator=g), torch.randn(Nd, 128, generator=g)
    r = torch.randn(max(E, 1), 128, generator=g)[:E]
    src = torch.randint(0, Ns, (E,), generator=g)
    dst = torch.sort(torch.randint(0, max(Nd - 1, 1), (E,), generator=g))[0]
    ref = orc.attention_layer(Wt, prefix, SMALL_SPEC, xs, xd, r, src, dst, bip).numpy()
    eoff = np.zeros(Nd + 1, np.int64)
    np.add.at(eoff, dst.numpy() + 1, 1)
    eoff = np.cumsum(eoff)
    rt = _rnorm(r) if E else np.zeros((1, 128), np.float32)
    li = small_engine.layer_index(group, int(prefix[-1]))
    for T in (11, 2, 4, 16):   # rows per workgroup of k_attn_chain; 11 = 1 row on 4 waves, two workgroups per CU; 16 = split layer (k_node + k_edge_small) when degree <= 128
        out = small_engine.test_attn(li, xs.numpy(), xd.numpy(), rt, eoff, src.numpy(), T)
        assert err(out, ref) < 2e-5, (T, err(out, ref))


def test_neighbour_sets_bit_exact(small_engine):
    """knn / radius edge SETS must equal the oracle's exactly (integer work: no tolerance)."""
    spec = SMALL_SPEC
    scene = synth.make_scene(spec, 24, 160, batch=3, seed=5, goal=True, ragged=True)
    small_engine.set_scene(scene)
    small_engine.rollout()
    tt = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(dt)
    mm, om = tt(scene["map_mask"], torch.bool).any(-1), tt(scene["prompt_mask"], torch.bool)
    m_pos, o_pos = tt(scene["map_pos"])[mm], tt(scene["obs_pos"])[om]
    mb, ob = orc._flat_batch_idx(mm), orc._flat_batch_idx(om)
    s_pos, sb = torch.cat([m_pos, o_pos]), torch.cat([mb, ob])
    Mv = int(mm.sum())
    d, s = orc.knn_edges(s_pos, sb, s_pos, sb, spec.scene_knn)
    es, ed, _ = small_engine.get_edges(1)
    assert set(zip(d.tolist(), s.tolist())) == set(zip(ed.tolist(), es.tolist()))
    d, s = orc.knn_edges(o_pos, ob, o_pos, ob, spec.agent_knn)
    es, ed, _ = small_engine.get_edges(0)
    assert set(zip(d.tolist(), (s + Mv).tolist())) == set(zip(ed.tolist(), es.tolist()))
    d, s = orc.radius_edges(o_pos, ob, o_pos, ob, spec.dec_prompt_radius, spec.dec_max_neigh, drop_self=True)
    es, ed, _ = small_engine.get_edges(2)
    assert list(zip(d.tolist(), (s + Mv).tolist())) == list(zip(ed.tolist(), es.tolist()))   # also the ORDER
    d, s = orc.radius_edges(s_pos, sb, o_pos, ob, spec.dec_scene_radius, spec.dec_max_neigh)
    es, ed, _ = small_engine.get_edges(3)
    assert list(zip(d.tolist(), s.tolist())) == list(zip(ed.tolist(), es.tolist()))


def test_radius_cap_truncation_index_order():
    """max_num_neighbors smaller than the neighbourhood: the FIRST cap candidates in index order
    survive (torch_cluster CUDA semantics), for radius and radius_graph(loop=False)."""
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC.replace(dec_max_neigh=8, pol_max_neigh=5)
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 20, 64, batch=2, seed=9, square=60.0)
    eng = Engine(spec, w)
    eng.set_scene(scene)
    eng.rollout()
    with torch.no_grad():
        o = orc.rollout(w, spec, scene)
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
    ec = eng.get("edge_counts")
    assert int(ec[2]) == o["edges"]["p2p"] and int(ec[3]) == o["edges"]["s2p"]
    assert int(ec[4]) == o["step_edges"][-1]["a2p"] and int(ec[5]) == o["step_edges"][-1]["m2p"]
    assert int(ec[3]) == 40 * 8 and int(ec[5]) <= 40 * 5
    A = eng.num_agents
    assert err(eng.get("motion_pred")[0], o64["motion_pred"][:A].numpy()) < TOL
    eng.close()


# ------------------------------------------------------------------ full rollouts vs golden fixtures
# (the four full-size REPORT_ONLY fixtures of round 5 have their own test with the cut-agent gate: tests/test_round5_gpu.py)
@pytest.mark.parametrize("name", [n for n in FULL_CASES if n not in REPORT_ONLY])
def test_rollout_vs_reference_fixture(name):
    from prosim_amd.engine import Engine
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    w = weights.init_weights(spec, wseed)
    scene = synth.make_scene(spec, **kw)
    assert digest(scene) == str(g["scene_digest"]) and digest(w) == str(g["weight_digest"])
    if "mode_choice" in g.files:      # TOP_K > 1: the reference's own mode draws (ps_set_mode_choice)
        scene["mode_choice"] = g["mode_choice"]
    if "action_noise" in g.files:     # RANDOM_NOISE_STD > 0: the reference's own noise draws, replayed
        scene["action_noise"] = g["action_noise"]
    eng = Engine(spec, w)
    eng.set_scene(scene)
    eng.rollout()
    pol = eng.policy_rows                                    # the reference returns policy agents only
    A = int(pol.sum())
    assert A == eng.num_policy_agents and (A < eng.num_agents) == ("replay" in kw)
    mp = eng.get("motion_pred")[:, pol].reshape(-1, *g["motion_pred"].shape[1:])
    assert err(mp[:A], g["motion_pred"][:A]) < TOL          # replan 0: open loop
    if spec.use_goal_pred_loss:
        assert err(eng.get("reconst_pred")[pol], g["reconst_pred"]) < 1e-5
    else:   # (no pred_mlp in the model: the result does not exist)
        with pytest.raises(RuntimeError):
            eng.get("reconst_pred")
    floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
    assert err(eng.padded("traj"), g["traj"]) < 3 * floor["traj"] + TOL
    assert err(eng.padded("vel"), g["vel"]) < 3 * floor["vel"] + TOL
    assert err(mp, g["motion_pred"]) < 3 * floor["motion_pred"] + TOL
    eng.close()


@pytest.mark.parametrize("cfg_idx,batch,seed", [(1, None, 0), (2, None, 0), (3, 2, 0), (3, 2, 1), (4, None, 0)])
def test_baseline_configs_vs_oracle(demo_engine, cfg_idx, batch, seed):
    """BASELINE.json configs at full size (config 3's 8-scene batch cut to 2 scenes to keep the
    oracle quick): stage outputs and the closed loop against the fp64 restatement, 1e-4 absolute.
    Config 3 runs on seed 0 AND seed 1: seed 0 puts a map polyline 1.4e-6 rad from the +-pi cut of one agent's frame at
    replan 2 -- whichever side a given fp32 rounding picks, that agent and its neighbours leave the fp64 trajectory by
    1e-3 (the flip class of the bar below); both land in the parity table with their flip rates."""
    spec = DEMO_SPEC
    scene = synth.baseline_scene(spec, cfg_idx, seed=seed, batch=batch)
    w = weights.init_weights(spec, 0)
    from oracle_cache import oracle64
    o64 = oracle64(f"baseline_cfg{cfg_idx}_b{batch}_s{seed}", spec, w, scene, collect=True)   # (tests/golden/oracle_cache, digest-checked)
    eng = demo_engine
    eng.set_scene(scene)
    eng.encode_scene()
    assert err(eng.get("scene_tokens"), o64["trace"]["scene_tokens"].numpy()) < TOL
    eng.generate_policy()
    pm = torch.from_numpy(scene["prompt_mask"].astype(bool))
    assert err(eng.get("policy_emd"), o64["policy_emd"][pm].numpy()) < 2 * TOL      # |emd| ~ 30
    ec = eng.get("edge_counts")
    assert [int(ec[i]) for i in range(4)] == [o64["edges"][k] for k in ("a2a", "s2s", "p2p", "s2p")]
    eng.reset_rollout()
    A = eng.num_agents
    for t in range(spec.n_replans):
        eng.policy_step(t)
    mp = eng.get("motion_pred")
    assert err(mp[0], o64["motion_pred"][:A].numpy()) < TOL
    # Closed loop, per agent.  The reference's math has branch cuts (wrap_angle / atan2 at +-pi feed
    # NON-periodic Fourier features, fourier_embedding.py:63-78), so an fp32 run -- the reference's
    # own included -- occasionally flips one edge feature of one agent and lands ~5e-4 away for a
    # replan (DESIGN.md "branch cuts").  Bar: >= 98 % of agents within 1e-4, nobody beyond 5e-3, median at the fp32
    # noise level.
    d_traj = np.abs(eng.padded("traj") - o64["traj"].numpy())[scene["prompt_mask"].astype(bool)].reshape(A, -1).max(1)
    d_vel = np.abs(eng.padded("vel") - o64["vel"].numpy())[scene["prompt_mask"].astype(bool)].reshape(A, -1).max(1)
    d_mp = np.abs(mp - o64["motion_pred"].numpy().reshape(mp.shape)).transpose(1, 0, 2, 3, 4).reshape(A, -1).max(1)
    print(f"cfg{cfg_idx}: per-agent max err  traj median {np.median(d_traj):.2e} max {d_traj.max():.2e} | vel max "
          f"{d_vel.max():.2e} | motion_pred max {d_mp.max():.2e} | agents within 1e-4: {(d_traj < TOL).mean():.3f}")
    from parity_table import per_agent, record, closed_loop_gate
    record(f"baseline_configs/cfg{cfg_idx}_seed{seed}", replan0_max=err(mp[0], o64["motion_pred"][:A].numpy()), **per_agent(d_traj))
    for d in (d_traj, d_vel, d_mp):
        closed_loop_gate(f"baseline_configs/cfg{cfg_idx}_seed{seed}", d)


def test_open_loop_policy_step_from_oracle_state(demo_engine):
    """policy.forward parity per replan, teacher-forced: load the oracle's trajectory state before
    replan t and run that replan only -- no closed-loop amplification, so 1e-4 absolute."""
    spec = DEMO_SPEC
    scene = synth.baseline_scene(spec, 1, seed=4)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o = orc.rollout(w, spec, scene, dtype=torch.float64)
    eng = demo_engine
    eng.set_scene(scene)
    eng.encode_scene()
    eng.generate_policy()
    A, H = eng.num_agents, spec.hist_steps
    hist_t = np.nan_to_num(scene["obs_input"][0, :, :, :4])
    hist_v = np.nan_to_num(scene["obs_input"][0, :, :, 4:6])
    full_t = np.concatenate([hist_t, o["traj"][0].numpy()], 1).astype(np.float32)
    full_v = np.concatenate([hist_v, o["vel"][0].numpy()], 1).astype(np.float32)
    for t in (0, 3, 7):
        n = H + t * spec.replan_freq
        eng.set_state(full_t[:, :n], full_v[:, :n])
        eng.policy_step(t)
        assert err(eng.get("motion_pred")[t], o["motion_pred"][t * A:(t + 1) * A].numpy()) < TOL


# ------------------------------------------------------------------ size-independent properties
def test_batch_independence_and_determinism(demo_engine):
    """Scenes never mix (all graph ops are batch-segmented, e.g. act_decoder.py:250): a 3-scene
    batch must reproduce each scene rolled out alone; and a rerun must be bit-identical."""
    spec = DEMO_SPEC
    scene = synth.make_scene(spec, 48, 256, batch=3, seed=11, goal=True, ragged=True)
    eng = demo_engine
    eng.set_scene(scene)
    eng.rollout()
    traj = eng.padded("traj")
    mp0 = eng.padded("policy_emd")
    eng.rollout()
    assert np.array_equal(traj, eng.padded("traj"))
    for b in range(3):
        one = {k: (v[b:b + 1] if not isinstance(v, dict) else {kk: {k3: v3[b:b + 1] for k3, v3 in vv.items()} for kk, vv in v.items()})
               for k, v in scene.items()}
        eng.set_scene(one)
        eng.rollout()
        # same maths, different tile shapes (rows per workgroup, waves per destination): open-loop
        # quantities agree to rounding, the closed loop to the amplified-rounding bound
        # (a radius-graph membership flip or a +-pi branch cut can send ONE closed loop down another
        # branch -- the fp32 oracle does the same against the fp64 one on this very scene -- so the
        # closed-loop check is per agent: >= 90 % of the agents within 1e-3)
        assert err(eng.padded("policy_emd")[0], mp0[b]) < 1e-4
        n_b = int(scene["prompt_mask"][b].sum())
        d = np.abs(eng.padded("traj")[0, :n_b] - traj[b, :n_b]).reshape(n_b, -1).max(1)
        assert (d < 1e-3).mean() >= 0.9, d
    assert np.isfinite(traj).all()
    t = traj[scene["prompt_mask"].astype(bool)]
    assert err(t[..., 2] ** 2 + t[..., 3] ** 2, 1.0) < 1e-5   # (sin, cos) stays on the unit circle


def test_degenerate_scenes(small_engine):
    """single agent, single polyline with one valid point, agents far from any map token."""
    spec = SMALL_SPEC
    scene = synth.make_scene(spec, 1, 1, batch=1, seed=2, points=1)
    w = weights.init_weights(spec, 0)
    small_engine.set_scene(scene)
    small_engine.rollout()
    with torch.no_grad():
        o = orc.rollout(w, spec, scene, dtype=torch.float64)
    assert err(small_engine.get("motion_pred")[0], o["motion_pred"][:1].numpy()) < TOL
    scene = synth.make_scene(spec, 6, 10, batch=1, seed=3)
    scene["map_pos"] += 5000.0          # no m2p edge for anyone: agg = 0 path (attention_layer.py)
    small_engine.set_scene(scene)
    small_engine.rollout()
    with torch.no_grad():
        o = orc.rollout(w, spec, scene, dtype=torch.float64)
    assert int(small_engine.get("edge_counts")[5]) == 0 == o["step_edges"][-1]["m2p"]
    assert err(small_engine.get("motion_pred")[0], o["motion_pred"][:6].numpy()) < TOL


def test_errors_are_loud(small_engine):
    spec = SMALL_SPEC
    scene = synth.make_scene(spec, 4, 8, batch=1, seed=0)
    bad = dict(scene)
    bad["obs_mask"] = scene["obs_mask"].copy()
    bad["obs_mask"][0, 0] = False         # a policy agent (prompt present) that is not observed
    with pytest.raises(RuntimeError, match="policy agent must be observed"):
        small_engine.set_scene(bad)
    bad = dict(scene)
    bad["prompt_mask"] = np.zeros_like(scene["prompt_mask"])   # observed agents only: nothing to simulate
    with pytest.raises(RuntimeError, match="no policy agent"):
        small_engine.set_scene(bad)
    bad = dict(scene)
    bad["agent_type"] = scene["agent_type"] * 0
    with pytest.raises(RuntimeError, match="agent_type"):
        small_engine.set_scene(bad)


def test_future_obs_frames_and_conditions_subset(small_engine):
    """fut_obs[t] supplies the static observation columns (extent, type, time one-hot) of replans 1..R-1
    (dataset/format_utils.py:667-687; columns 0..7 are overwritten by step_env, traj_sam.py:266-270), and
    conditions may cover only some agents (condition_utils.py masks)."""
    spec = SMALL_SPEC
    scene = synth.make_scene(spec, 10, 48, batch=2, seed=21, goal=True, tags=True, ragged=True)
    rng = np.random.RandomState(0)
    R = spec.n_replans
    fut = np.repeat(np.nan_to_num(scene["obs_input"])[None], R - 1, axis=0).copy()
    fut[..., 8:10] += rng.uniform(-0.2, 0.2, fut[..., 8:10].shape).astype(np.float32)    # extents drift per frame
    scene["fut_obs_input"] = fut
    w = weights.init_weights(spec, 0)
    small_engine.set_scene(scene)
    small_engine.rollout()
    with torch.no_grad():
        o = orc.rollout(w, spec, scene, dtype=torch.float64)
        o_nofut = orc.rollout(w, spec, {k: v for k, v in scene.items() if k != "fut_obs_input"}, dtype=torch.float64)
    A = small_engine.num_agents
    mp = small_engine.get("motion_pred")
    assert err(mp[0], o["motion_pred"][:A].numpy()) < TOL
    assert err(mp[1], o["motion_pred"][A:2 * A].numpy()) < 2 * TOL           # first replan that sees a fut frame
    assert err(o["motion_pred"][A:2 * A].numpy(), o_nofut["motion_pred"][A:2 * A].numpy()) > 1e-3   # the frames matter
    d = np.abs(small_engine.padded("traj") - o["traj"].numpy())[scene["prompt_mask"].astype(bool)].reshape(A, -1).max(1)
    assert (d < 1e-3).mean() >= 0.9


def test_drag_point_conditions_vs_oracle(small_engine):
    """DragPointEncoder (condition_encoders.py:152-191): a PointNet over the non-NaN [x, y] points of each drag, pooled
    with the goal / tag entries of the same agent (condition_attns.py:114-188).  Checked on the generator output
    (policy_emd, fp64 oracle) with drag points alone, with all three types, and after clearing them."""
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    full = synth.make_scene(spec, 24, 64, batch=2, seed=33, goal=True, tags=True, drag=True, ragged=True)
    pm = full["prompt_mask"].astype(bool)
    assert np.isnan(full["cond"]["drag_point"]["input"]).any()             # absent points are NaN, as the dataset makes them

    def emd_of(scene):
        small_engine.set_scene(scene)
        small_engine.encode_scene()
        small_engine.generate_policy()
        with torch.no_grad():
            o = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
        return small_engine.padded("policy_emd")[pm], o["policy_emd"].numpy()[pm]

    only_drag = dict(full, cond={"drag_point": full["cond"]["drag_point"]})
    none = {k: v for k, v in full.items() if k != "cond"}
    got_d, want_d = emd_of(only_drag)
    assert err(got_d, want_d) < TOL
    got_f, want_f = emd_of(full)
    assert err(got_f, want_f) < TOL
    got_n, want_n = emd_of(none)                                            # set_scene without conditions clears them
    assert err(got_n, want_n) < TOL
    assert err(want_d, want_n) > 1e-2 and err(want_f, want_d) > 1e-2        # and every type moves the result
    # a drag whose points are all NaN is still an (all-zero) entry of its agent when its mask is set (:180-182)
    hole = dict(only_drag, cond={"drag_point": {k: v.copy() for k, v in full["cond"]["drag_point"].items()}})
    b0, c0 = np.argwhere(hole["cond"]["drag_point"]["mask"])[0]
    hole["cond"]["drag_point"]["input"][b0, c0] = np.nan
    got_h, want_h = emd_of(hole)
    assert err(got_h, want_h) < TOL
    # more than 32 points per drag is outside the PointNet kernel's tile
    long = dict(only_drag, cond={"drag_point": dict(full["cond"]["drag_point"], input=np.zeros((2, 24, 40, 2), np.float32))})
    with pytest.raises(RuntimeError, match="1..32 points"):
        small_engine.set_scene(long)
    # an engine built for a checkpoint without the drag-point encoder refuses drag conditions
    from prosim_amd.engine import Engine
    spec0 = spec.replace(drag_mlp_layers=0)
    w0 = weights.init_weights(spec0, 0)
    assert not any("drag_point" in k for k in w0) and all(np.array_equal(w0[k], w[k]) for k in w0)
    eng0 = Engine(spec0, w0)
    try:
        with pytest.raises(RuntimeError, match="without the drag-point encoder"):
            eng0.set_scene(only_drag)
        eng0.set_scene(dict(full, cond={k: v for k, v in full["cond"].items() if k != "drag_point"}))
    finally:
        eng0.close()


def test_demo_dataset_scene_config0():
    """BASELINE configs[0]: a demo_dataset scene (real Waymo tracks from the reference's sample cache, formatted by
    prosim_amd/formatting.py; lanes drawn along the driven paths), 16 agents, 20-step unconditional rollout.
    Ragged real histories (agents that appear mid-window) against the fp64 oracle."""
    from prosim_amd import formatting as fmt
    from prosim_amd.engine import Engine
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_scene_0_agent_table.npz"))
    tracks = fmt.tracks_from_table({k: g[k] for k in g.files if k != "origin"})
    spec = DEMO_SPEC.replace(max_steps=20)
    t0 = 10
    present = np.isfinite(tracks["x"][:, t0])
    late = present & ~np.isfinite(tracks["x"][:, 0])         # agents that appeared inside the history window first
    order = list(np.nonzero(late)[0]) + list(np.nonzero(present & ~late)[0])
    scene = fmt.scene_from_tracks(spec, tracks, t0, agents=order, max_agents=16)
    scene.pop("agent_ids")
    assert (~scene["obs_mask"][0, :, 0, :8]).any()          # some histories really are incomplete
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
    eng = Engine(spec, w)
    eng.set_scene(scene)
    eng.encode_scene()
    assert err(eng.get("scene_tokens"), o64["trace"]["scene_tokens"].numpy()) < TOL
    eng.rollout()
    A = eng.num_agents
    assert A == 16
    mp = eng.get("motion_pred")
    assert err(mp[0], o64["motion_pred"][:A].numpy()) < TOL
    d = np.abs(eng.padded("traj") - o64["traj"].numpy())[0].reshape(A, -1).max(1)
    assert (d < TOL).mean() >= 0.9 and d.max() < 5e-3, d
    eng.close()


@pytest.mark.parametrize("demo_scene", ["scene_1", "scene_0"])
def test_demo_dataset_real_lanes_config0(demo_scene):
    """BASELINE configs[0] on the scene's REAL lanes: demo scene_1 (agent table + the cache's VectorMap protobuf decoded by
    prosim_amd/vecmap.py, chunked and framed as data_utils.py:156-255 / format_utils.py:150-263 do), centred on the ego,
    16 agents (vehicles, a pedestrian, bicycles: the cache's own types), 20-step unconditional rollout.  Replan-0 predictions
    against the fp64 oracle to 1e-4, trajectories to the fp32 floor of the scene.
    The scene TOKENS are compared with both oracles: two antiparallel lane chunks of this map have a relative heading of
    pi to the last bit, which wrap_angle (geometry.py:13-17) sends to -pi or +pi depending on the rounding of the two
    headings -- one edge whose Fourier features flip sign between an fp32 and an fp64 evaluation (which side the engine
    lands on depends on the frame the scene is given in); every token row must sit on one of the two sides."""
    from prosim_amd.engine import Engine
    from test_vecmap_cpu import demo_scene_real_lanes
    spec = DEMO_SPEC.replace(max_steps=20)
    # scene_0: 2048 map tokens (the MAX_POINTS cap), traffic-light records, >= 2048 scene tokens -> the split s2s layers
    scene = demo_scene_real_lanes(spec, scene=demo_scene)
    assert scene["map_input"].shape[1] > 300
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
        o32 = orc.rollout(w, spec, scene, dtype=torch.float32, collect=True)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.encode_scene()
        tok = eng.get("scene_tokens")
        e32 = np.abs(tok - o32["trace"]["scene_tokens"].numpy()).max(1)
        e64 = np.abs(tok - o64["trace"]["scene_tokens"].numpy()).max(1)
        assert np.minimum(e32, e64).max() < TOL            # every token row is the fp32 or the fp64 side of the cut
        assert (e64 < TOL).mean() > 0.8
        eng.rollout()
        A = eng.num_agents
        assert A == 16
        assert err(eng.get("motion_pred")[0], o64["motion_pred"][:A].numpy()) < TOL
        d = np.abs(eng.padded("traj") - o64["traj"].numpy())[0].reshape(A, -1).max(1)
        floor = float((o32["traj"].double() - o64["traj"]).abs().max())     # what fp32 arithmetic alone does on this scene
        assert d.max() < 3 * floor + TOL and np.median(d) < TOL, (d, floor)
    finally:
        eng.close()


def test_demo_dataset_full_rollout_with_real_log_replay():
    """The whole path on the reference's sample data: demo scene_1 at step 10 -- real lanes, real agent types, 12 policy
    agents, the 33 other agents of the scene replaying their real logs through the fut_obs frames (agents that leave and
    agents that enter), goal and drag-point prompts taken from the log, 80 steps -- against the oracle: replan 0 to 1e-4, trajectories to the scene's fp32 floor."""
    from prosim_amd.engine import Engine
    from test_vecmap_cpu import demo_rollout_batch
    spec = DEMO_SPEC
    scene, _, policy, replay = demo_rollout_batch(spec)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
        o32 = orc.rollout(w, spec, scene)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.rollout()
        assert eng.num_agents == len(policy) + len(replay) and eng.num_policy_agents == len(policy)
        pol = eng.policy_rows
        P = int(pol.sum())
        assert err(eng.get("motion_pred")[0][pol], o64["motion_pred"][:P].numpy()) < TOL
        pm = scene["prompt_mask"].astype(bool)
        d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(P, -1).max(1)
        floor = float((o32["traj"].double() - o64["traj"]).abs()[torch.from_numpy(pm)].max())
        assert d.max() < 3 * floor + TOL and np.median(d) < floor + TOL, (d, floor)
        # ... and the validation metric against the REAL log (pair_targets_from_tracks), on the device and through the oracle
        from oracle import metric_oracle as mo
        from prosim_amd import formatting as fmt
        from prosim_amd.distributed import reduce_pair_metrics, rows_to_slots
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_scene_1_agent_table.npz"))
        tr = fmt.tracks_from_table({k: g[k] for k in g.files if k != "origin"})
        gt = fmt.pair_targets_from_tracks(spec, tr, 10, policy + replay)
        slots, A, R = eng.row_slots, eng.num_agents, spec.n_replans
        tgt_rows = np.ascontiguousarray(gt["tgt"][0][:, slots])                  # [R, A, S, 5]
        mask_rows = np.ascontiguousarray(gt["mask"][0][:, slots] & pol[None])
        dev = torch.device("cuda", 0)
        t_tgt, t_mask = torch.from_numpy(tgt_rows).to(dev), torch.from_numpy(mask_rows.astype(np.uint8)).to(dev)
        out = torch.zeros(A, 10, device=dev)
        eng.pair_metric(out.data_ptr(), t_tgt.data_ptr(), t_mask.data_ptr())
        eng.sync()
        got = out.cpu().numpy()
        mp = eng.get("motion_pred")
        r_idx, a_idx = np.nonzero(mask_rows)
        o = mo.pair_motion_pred(torch.from_numpy(mp[r_idx, a_idx]), torch.ones(len(r_idx), 1), torch.from_numpy(tgt_rows.transpose(1, 0, 2, 3)[None].transpose(0, 2, 1, 3, 4).copy()),
                                torch.from_numpy(mask_rows[None].copy()), np.zeros_like(r_idx), r_idx, a_idx, spec.replan_freq)
        red = reduce_pair_metrics(rows_to_slots(torch.from_numpy(got), torch.from_numpy(slots), 1, len(slots)))
        for k in ("ade", "fde", "min_ade", "min_fde", "rollout_ade"):
            assert abs(red[k] - float(o[k])) < 1e-4 * max(1.0, abs(float(o[k]))), (k, red[k], float(o[k]))
        assert red["ade"] > 0.5                                                   # random weights against a real log: far off, finite
    finally:
        eng.close()


def test_split_s2s_layers_on_a_ragged_batch(demo_engine):
    """>= 2048 scene tokens switch the s2s layers to the split launches (k_node + k_edge_small): a ragged 2-scene
    batch (token counts that are no multiple of the 16-row tiles, polylines with few valid points, incomplete
    histories) against the fp64 oracle's scene tokens and generator output."""
    spec = DEMO_SPEC
    scene = synth.make_scene(spec, 160, 1100, batch=2, seed=21, goal=True, ragged=True)
    w = weights.init_weights(spec, 0)
    from oracle_cache import oracle64
    o64 = oracle64("split_s2s_ragged", spec, w, scene, collect=True)
    eng = demo_engine
    eng.set_scene(scene)
    assert eng.num_agents + eng.num_map_tokens >= 2048
    eng.encode_scene()
    assert err(eng.get("scene_tokens"), o64["trace"]["scene_tokens"].numpy()) < TOL
    eng.generate_policy()
    pm = torch.from_numpy(scene["prompt_mask"].astype(bool))
    assert err(eng.get("policy_emd"), o64["policy_emd"][pm].numpy()) < 2 * TOL


def test_log_replay_agents_vs_oracle(demo_engine):
    """Policy agents as a subset of the observed agents at demo-model size: a third of the observed agents replay a
    log (their token is re-encoded from the logged observation at the logged pose every replan, some drop out of the
    log), goal + action-tag prompts on the policy agents.  Per-agent closed-loop bar against the fp64 oracle."""
    spec = DEMO_SPEC
    scene = synth.make_scene(spec, 64, 384, batch=2, seed=31, goal=True, tags=True, ragged=True, replay=0.35)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
    eng = demo_engine
    eng.set_scene(scene)
    pol = eng.policy_rows
    A = int(pol.sum())
    assert 0 < A < eng.num_agents
    eng.encode_scene()
    assert err(eng.get("scene_tokens"), o64["trace"]["scene_tokens"].numpy()) < TOL
    eng.generate_policy()
    pm = torch.from_numpy(scene["prompt_mask"].astype(bool))
    assert err(eng.get("policy_emd")[pol], o64["policy_emd"][pm].numpy()) < 2 * TOL
    eng.rollout()
    mp = eng.get("motion_pred")[:, pol]
    assert err(mp[0], o64["motion_pred"][:A].numpy()) < TOL
    # closed loop: relative to what fp32 itself loses on this (ragged, dense) scene -- the oracle's own fp32 run
    with torch.no_grad():
        o32 = orc.rollout(w, spec, scene)
    floor = float(np.abs(o32["traj"].numpy() - o64["traj"].numpy()).max())
    d = np.abs(eng.padded("traj") - o64["traj"].numpy())[scene["prompt_mask"].astype(bool)].reshape(A, -1).max(1)
    assert d.max() < 3 * floor + TOL and (d < TOL).mean() >= 0.8, (floor, d)
    assert np.abs(eng.padded("traj")[~scene["prompt_mask"].astype(bool)]).max() == 0      # log-replay slots stay empty


@pytest.mark.parametrize("fusion,attn", [("mlp", False), ("replace", True), ("mlp", True)])
def test_obs_update_variants_vs_oracle(fusion, attn):
    """MODEL.OBS_UPDATE (attn_fusion.py:136-203, default.py:499-501), every agent policy-controlled: FUSION 'mlp' folds
    the re-encoded observation into the previous token, ATTN_UPDATE re-runs the encoder's a2a / s2s(map -> agent)
    layers at the new poses.  Replan 1 is the first step that sees the update: open-loop bar there, closed-loop bar
    relative to the fp32 floor on the trajectories."""
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC.replace(obs_fusion=fusion, obs_attn_update=attn)
    w = weights.init_weights(spec, 0)
    base = weights.init_weights(SMALL_SPEC, 0)
    assert all(np.array_equal(w[k], base[k]) for k in base)                   # the variants only ADD tensors
    scene = synth.make_scene(spec, 24, 96, batch=2, seed=41, goal=True, ragged=True, clustered=True)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
        o32 = orc.rollout(w, spec, scene)
        plain = orc.rollout(w, SMALL_SPEC, scene, dtype=torch.float64)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.rollout()
        A = eng.num_agents
        mp = eng.get("motion_pred")
        assert err(mp[0], o64["motion_pred"][:A].numpy()) < TOL
        assert err(mp[1], o64["motion_pred"][A:2 * A].numpy()) < 2 * TOL
        assert err(o64["motion_pred"][A:2 * A].numpy(), plain["motion_pred"][A:2 * A].numpy()) > 1e-2   # the variant matters
        floor = float(np.abs(o32["traj"].numpy() - o64["traj"].numpy()).max())
        pm = scene["prompt_mask"].astype(bool)
        d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
        assert d.max() < 3 * floor + TOL, (floor, d.max())
    finally:
        eng.close()
    with pytest.raises(ValueError, match="FUSION"):
        Engine(SMALL_SPEC.replace(obs_fusion="sum"), base)


def test_maximum_scene_size(small_engine):
    """The engine's per-scene limits at once: 2560 tokens (512 agents + 2048 polylines, the kNN kernel's candidate
    registers), 32 points per polyline (the PointNet tile).  One more token is refused loudly."""
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 512, 2048, batch=1, seed=77, goal=True, points=32, square=400.0)
    from oracle_cache import oracle64
    o64 = oracle64("maximum_scene_size", spec, w, scene)
    small_engine.set_scene(scene)
    small_engine.rollout()
    A = small_engine.num_agents
    assert A == 512 and small_engine.num_map_tokens == 2048
    assert err(small_engine.get("motion_pred")[0], o64["motion_pred"][:A].numpy()) < TOL
    pm = scene["prompt_mask"].astype(bool)
    d = np.abs(small_engine.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
    assert (d < TOL).mean() >= 0.97 and np.median(d) < 2e-5, (d.max(), (d < TOL).mean())   # isolated +-pi flips aside
    with pytest.raises(RuntimeError, match="2560 tokens"):
        small_engine.set_scene(synth.make_scene(spec, 513, 2048, batch=1, seed=1))
    with pytest.raises(RuntimeError, match="P <= 32"):
        small_engine.set_scene(synth.make_scene(spec, 4, 8, batch=1, seed=1, points=33))


def test_condition_type_present_but_every_entry_masked(small_engine):
    """A condition TYPE that is present in the batch makes the reference run the condition layers over every policy
    agent (without edges they still add their node update to the embedding), even when each entry is masked off:
    condition_transformer/base.py:43-49, condition_encoders.py:106-111, condition_attns.py:203-228.  Found by the
    randomised sweep (one agent whose only tag entry was masked)."""
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 6, 24, batch=2, seed=52, goal=True, tags=True, drag=True)
    pm = scene["prompt_mask"].astype(bool)

    def emd_of(sc):
        small_engine.set_scene(sc)
        small_engine.encode_scene()
        small_engine.generate_policy()
        with torch.no_grad():
            o = orc.rollout(w, spec, sc, dtype=torch.float64)
        return small_engine.padded("policy_emd")[pm], o["policy_emd"].numpy()[pm]

    none = {k: v for k, v in scene.items() if k != "cond"}
    got_n, want_n = emd_of(none)
    for keep in (("goal",), ("v_action_tag",), ("drag_point",), ("goal", "v_action_tag", "drag_point")):
        cond = {k: dict(scene["cond"][k], mask=np.zeros_like(scene["cond"][k]["mask"])) for k in keep}
        got, want = emd_of(dict(scene, cond=cond))
        assert err(got, want) < TOL, keep
        assert err(want, want_n) > 1e-2, keep            # the layers ran although no entry is valid
    assert err(got_n, want_n) < TOL
    # tag rows whose id is no V_Action tag (-1 = invalid) do not make the type present (condition_encoders.py:106-111)
    tags = dict(scene["cond"]["v_action_tag"], input=scene["cond"]["v_action_tag"]["input"].copy())
    tags["input"][..., 0] = -1
    got, want = emd_of(dict(scene, cond={"v_action_tag": tags}))
    assert err(got, want) < TOL and err(want, want_n) < 1e-9


def test_throughput_mode_rows_per_workgroup(small_engine):
    """ps_set_chain_rows(4) (throughput mode for pipelined rollouts) changes the tiling of the fused attention launches
    with >= 512 rows, not the results: same parity against the fp64 oracle, and fp32-summation-order close to the
    latency-mode rollout of the same scene."""
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 150, 96, batch=4, seed=71, goal=True, tags=True)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
    outs = {}
    try:
        for rows in (0, 4, 2):
            small_engine.set_chain_rows(rows)
            small_engine.set_scene(scene)
            small_engine.rollout()
            A = small_engine.num_agents
            assert A == 600
            mp = small_engine.get("motion_pred")
            assert err(mp[0], o64["motion_pred"][:A].numpy()) < TOL, rows
            pm = scene["prompt_mask"].astype(bool)
            d = np.abs(small_engine.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
            assert (d < TOL).mean() >= 0.97 and np.median(d) < 3e-5, (rows, d.max())
            outs[rows] = mp
        # 0 = the engine's own choice: k_chain16 above 256 rows since the end of round 6 (the 2- and 4-row builds of k_attn_chain until then), another
        # kernel, another summation order -- fp32-close, like the two explicit tilings to each other
        assert err(outs[0][0], outs[2][0]) < 1e-5
        assert err(outs[4][0], outs[2][0]) < 1e-5
        with pytest.raises(RuntimeError, match="0 .auto., 1, 2, 4, or 8..16"):
            small_engine.set_chain_rows(3)
    finally:
        small_engine.set_chain_rows(0)
