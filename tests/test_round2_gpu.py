"""Round-2 GPU tests: the exact bench.py workload in every engine mode, the no-truncation variants of the BASELINE
configs, both fused-chain kernels against each other, the reference's rollout metric on the device, and the 2-rank
bench run.  Every closed-loop test writes its measured errors into the parity table (tests/parity_table.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from oracle import prosim_oracle as orc
from oracle import metric_oracle as mo
from parity_table import per_agent, record, closed_loop_gate
from golden_cases import make_pair_metric_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def cat_scenes(parts):
    return {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
                {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]})
            for k in parts[0]}


def closed_loop_errors(eng, scene, o64):
    pm = scene["prompt_mask"].astype(bool)
    A = eng.num_agents
    mp = eng.get("motion_pred")
    e0 = err(mp[0], o64["motion_pred"][:A].numpy())
    d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(A, -1).max(1)
    return e0, d


# ------------------------------------------------------------------ the timed configuration, every mode
@pytest.fixture(scope="module")
def bench_workload():
    """8 x BASELINE configs[2] scenes, seeds 0..7 (= configs[3]'s per-GPU share, what bench.py times), the fp64 oracle and
    the fp32 oracle's own distance from it (the fp32 floor)."""
    from oracle_cache import oracle64
    spec = DEMO_SPEC
    w = weights.init_weights(spec, 0)
    scene = cat_scenes([synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)])
    o64 = oracle64("bench_workload", spec, w, scene, floor=True)          # (tests/golden/oracle_cache, digest-checked)
    floor = o64["fp32_floor"]
    return spec, w, scene, o64, floor


@pytest.mark.parametrize("mode,impl,rows", [("latency", 0, 0), ("throughput", 0, 16), ("k_attn_chain_rows4", 1, 4),
                                            ("k_chain16_rows4", 2, 4), ("k_chain16_rows12", 2, 12)])
def test_bench_workload_parity(bench_workload, mode, impl, rows):
    """The configuration the headline number is measured on, in the engine modes bench.py uses (latency: one rollout on
    the GPU; throughput: ps_set_chain_rows(16)) and in the other kernel / tiling choices.  Open loop to 1e-4; closed
    loop per agent (DESIGN.md 'branch cuts': an fp32 run, the reference's included, can send single agents down another
    branch at a +-pi crossing -- the fp32 oracle's own numbers are in the table beside ours)."""
    from prosim_amd.engine import Engine
    spec, w, scene, o64, floor = bench_workload
    eng = Engine(spec, w)
    try:
        eng.set_chain_impl(impl)
        eng.set_chain_rows(rows)
        eng.set_scene(scene)
        eng.rollout()
        e0, d = closed_loop_errors(eng, scene, o64)
    finally:
        eng.close()
    record(f"bench_workload/{mode}", replan0_max=e0, fp32_oracle=per_agent(floor), **per_agent(d))
    print(f"bench workload [{mode}]: replan-0 {e0:.2e} | per agent median {np.median(d):.2e} p99 {np.percentile(d, 99):.2e} "
          f"max {d.max():.2e} within 1e-4: {(d < TOL).mean():.4f} | fp32 oracle: median {np.median(floor):.2e} max {floor.max():.2e} "
          f"within 1e-4: {(floor < TOL).mean():.4f}")
    assert e0 < TOL
    closed_loop_gate(f"bench_workload/{mode}", d)


# ------------------------------------------------------------------ no-truncation variants
@pytest.mark.parametrize("cfg_idx", [1, 2, 4])
def test_no_truncation_variant(cfg_idx):
    """Neighbour caps >= every candidate count: the result no longer depends on which neighbours a truncating
    torch_cluster would have kept (index order on CUDA, KD-tree order on CPU) -- the one place where the reference's own
    two back-ends may differ."""
    from prosim_amd.engine import Engine
    kw = synth.BASELINE_CONFIGS[cfg_idx]
    cap = kw["n_agents"] + kw["n_polylines"]
    spec = DEMO_SPEC.replace(dec_max_neigh=cap, pol_max_neigh=max(DEMO_SPEC.pol_max_neigh, min(cap, 2047)))
    w = weights.init_weights(spec, 0)
    scene = synth.baseline_scene(spec, cfg_idx, seed=0)
    from oracle_cache import oracle64
    o64 = oracle64(f"no_truncation_cfg{cfg_idx}", spec, w, scene)
    n_pol = int(scene["prompt_mask"].sum())
    assert o64["edges"]["s2p"] <= n_pol * cap and o64["edges"]["p2p"] <= n_pol * cap        # (a cap of ALL tokens cannot truncate)
    assert all(se["a2p"] <= n_pol * spec.pol_max_neigh and se["m2p"] <= n_pol * spec.pol_max_neigh for se in o64["step_edges"])
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.rollout()
        ec = eng.get("edge_counts")
        assert int(ec[2]) == o64["edges"]["p2p"] and int(ec[3]) == o64["edges"]["s2p"]
        e0, d = closed_loop_errors(eng, scene, o64)
    finally:
        eng.close()
    record(f"no_truncation/cfg{cfg_idx}", replan0_max=e0, caps=cap, **per_agent(d))
    print(f"no-truncation cfg{cfg_idx}: replan-0 {e0:.2e} | per agent median {np.median(d):.2e} max {d.max():.2e} within 1e-4 {(d < TOL).mean():.3f}")
    assert e0 < TOL
    closed_loop_gate(f"no_truncation/cfg{cfg_idx}", d)


# ------------------------------------------------------------------ the two fused-chain kernels against each other
def test_chain_kernels_agree_on_ragged_batches():
    """k_chain16 (every rows-per-workgroup choice, incl. rows shared by 2 / 4 / 8 waves and the row queue) against
    k_attn_chain on ragged batches with log-replay agents, empty edge lists and conditions: one open-loop policy step."""
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    eng = Engine(spec, w)
    try:
        for kw in (dict(n_agents=24, n_polylines=160, batch=3, seed=5, goal=True, ragged=True),
                   dict(n_agents=37, n_polylines=90, batch=5, seed=8, goal=True, tags=True, ragged=True, replay=0.4),
                   dict(n_agents=1, n_polylines=1, batch=1, seed=2, points=1),
                   dict(n_agents=150, n_polylines=600, batch=2, seed=3, square=80.0)):
            scene = synth.make_scene(spec, **kw)
            ref = None
            for impl, rows in ((1, 0), (2, 0), (2, 1), (2, 2), (2, 4), (2, 8), (2, 11), (2, 16)):
                eng.set_chain_impl(impl)
                eng.set_chain_rows(rows)
                eng.set_scene(scene)
                eng.encode_scene(); eng.generate_policy(); eng.reset_rollout(); eng.policy_step(0)
                mp, fused = eng.get("motion_pred")[0], eng.get("fused")
                if ref is None:
                    ref = (mp, fused)
                assert np.isfinite(fused).all()
                assert err(mp, ref[0]) < 2e-5 and err(fused, ref[1]) < 2e-4, (kw, impl, rows, err(mp, ref[0]), err(fused, ref[1]))
    finally:
        eng.close()
    # a distance beyond fdiv16's range (> 10 km between an agent and a map token it still sees: radius widened) takes the
    # true-division path of the feature rows
    spec_far = SMALL_SPEC.replace(pol_map_radius=30000.0, pol_agent_radius=30000.0)
    wf = weights.init_weights(spec_far, 0)
    scene = synth.make_scene(spec_far, 6, 12, batch=1, seed=4)
    scene["map_pos"][0, :4] += 15000.0
    eng = Engine(spec_far, wf)
    try:
        outs = []
        for impl in (1, 2):
            eng.set_chain_impl(impl)
            eng.set_scene(scene)
            eng.encode_scene(); eng.generate_policy(); eng.reset_rollout(); eng.policy_step(0)
            outs.append(eng.get("motion_pred")[0])
        with torch.no_grad():   # (fp32 oracle: at 15 km the fp32 rounding of the Fourier arguments IS the function being matched)
            o = orc.rollout(wf, spec_far, scene)
        # the two kernels share every rounding up to the sin / cos themselves; the oracle's torch.norm may round a 15 km
        # distance one ulp (1 mm = 6e-3 rad at the highest frequency) away from sqrtf(dx*dx + dy*dy): a loose bar there
        assert err(outs[0], outs[1]) < 2e-5 and err(outs[1], o["motion_pred"][:6].numpy()) < 2e-3
    finally:
        eng.close()


# ------------------------------------------------------------------ the reference's rollout metric on the device
def test_pair_metric_device_vs_oracle():
    """ps_pair_metric against oracle/metric_oracle.py (itself pinned to the reference's PairMotionPred by
    tests/golden/ref_pair_metric.npz, test_oracle_golden.py) on the engine's own predictions and seeded targets with
    gaps; log-replay rows come back NaN."""
    from prosim_amd.engine import Engine
    from prosim_amd.distributed import reduce_pair_metrics, rows_to_slots
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 12, 64, batch=3, seed=9, goal=True, ragged=True, replay=0.3)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.rollout()
        eng.sync()
        A, R, S = eng.num_agents, spec.n_replans, spec.target_steps
        pol = eng.policy_rows
        mp = eng.get("motion_pred")                                           # [R, A, K=1, S, 5]
        B, N = scene["prompt_mask"].shape
        d = make_pair_metric_inputs(3, B=B, N=N, R=R, K=1, S=S)
        slots = eng.row_slots
        tgt_rows = d["tgt"].reshape(B, R, N, S, 5).transpose(1, 0, 2, 3, 4).reshape(R, B * N, S, 5)[:, slots]   # [R, A, S, 5]
        mask_rows = d["mask"].transpose(1, 0, 2).reshape(R, B * N)[:, slots] & pol[None]
        dev = torch.device("cuda", 0)
        t_tgt = torch.from_numpy(np.ascontiguousarray(tgt_rows)).to(dev)
        t_mask = torch.from_numpy(np.ascontiguousarray(mask_rows.astype(np.uint8))).to(dev)
        out = torch.zeros(A, 10, device=dev)
        eng.pair_metric(out.data_ptr(), t_tgt.data_ptr(), t_mask.data_ptr())
        eng.sync()
        got = out.cpu().numpy()
    finally:
        eng.close()
    assert np.isnan(got[~pol]).all() and np.isfinite(got[pol]).all()
    # the same through the oracle: one "scene" holding the A rows
    r_idx, a_idx = np.nonzero(mask_rows)
    o = mo.pair_motion_pred(torch.from_numpy(mp[r_idx, a_idx]), torch.ones(len(r_idx), 1), torch.from_numpy(tgt_rows.transpose(1, 0, 2, 3)[None].transpose(0, 2, 1, 3, 4).copy()),
                            torch.from_numpy(mask_rows[None].copy()), np.zeros_like(r_idx), r_idx, a_idx, spec.replan_freq)
    tot = np.nansum(got, axis=0)
    for i, k in enumerate(("ade", "fde", "min_ade", "min_fde")):
        assert abs(tot[i] / tot[4 + i] - float(o[k])) < 1e-5 * max(1.0, abs(float(o[k]))), (k, tot[i] / tot[4 + i], float(o[k]))
    want_ra, want_valid = o["agent_rollout_ade"][0].numpy(), o["agent_valid"][0].numpy()
    assert np.array_equal(got[pol, 9] > 0, want_valid[pol])
    assert err(got[pol, 8][want_valid[pol]], want_ra[pol][want_valid[pol]]) < 1e-4
    red = reduce_pair_metrics(rows_to_slots(torch.from_numpy(got), torch.from_numpy(slots), B, N))
    assert abs(red["rollout_ade"] - float(o["rollout_ade"])) < 1e-4 * max(1.0, float(o["rollout_ade"]))
    assert abs(red["ade"] - float(o["ade"])) < 1e-5 * max(1.0, float(o["ade"]))


# ------------------------------------------------------------------ optional model variants
def test_goal_heads_vs_reference_fixture():
    """MODEL.DECODER.GOAL_PRED (decoder/base.py:22-58): goal_prob / goal_point of the reference's own decoder."""
    from prosim_amd.engine import Engine
    from golden_cases import GOAL_CASE, digest
    name, spec, kw, wseed = GOAL_CASE
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"ref_standins_{name}.npz"))
    w = weights.init_weights(spec, wseed)
    scene = synth.make_scene(spec, **kw)
    assert digest(scene) == str(g["scene_digest"]) and digest(w) == str(g["weight_digest"])
    pm = scene["prompt_mask"].astype(bool)
    eng = Engine(spec, w)
    try:
        eng.set_scene({k: v for k, v in scene.items() if k != "cond"})     # the decoder output precedes the condition layers
        eng.encode_scene()
        eng.generate_policy()
        assert err(eng.padded("policy_emd")[pm], g["emd"][pm]) < TOL
        assert err(eng.padded("goal_prob")[pm], g["goal_prob"][pm]) < 1e-5
        assert err(eng.padded("goal_point")[pm], g["goal_point"][pm]) < 1e-5
    finally:
        eng.close()
    eng0 = Engine(SMALL_SPEC, weights.init_weights(SMALL_SPEC, 0))
    try:
        eng0.set_scene(scene)
        eng0.encode_scene(); eng0.generate_policy()
        with pytest.raises(RuntimeError, match="without goal heads"):
            eng0.get("goal_prob")
    finally:
        eng0.close()


def test_top_k_draw_replays_the_reference_stream():
    """ROLLOUT.POLICY.TOP_K = 3 through the registry-level model: ProSimHip draws the modes with the reference's own
    torch.topk / torch.randint calls, so the fixture's seed reproduces the fixture's draws -- and its trajectories."""
    from prosim_amd import modules
    from oracle import ref_batch as rh
    from golden_cases import FULL_CASES, SPECS, TOPK_SEED
    sname, kw, wseed = FULL_CASES["small_topk3_b2"]
    spec = SPECS[sname]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_standins_small_topk3_b2.npz"))
    w = weights.init_weights(spec, wseed)
    scene = synth.make_scene(spec, **kw)
    model = modules.ProSimHip(spec, w)
    try:
        batch = rh.make_batch(scene, spec)
        torch.manual_seed(int(g["torch_seed"]))
        out = model.forward(batch, "val")["motion_pred"]
        pm = scene["prompt_mask"].astype(bool)
        drawn = model._last_mode_choice
        assert np.array_equal(drawn[:, pm], g["mode_choice"][:, pm])
        floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
        B, N = pm.shape
        for b in range(B):
            for n in np.nonzero(pm[b])[0]:
                assert err(out["rollout_trajs"][f"{b}-a{n}"]["traj"].numpy(), g["traj"][b, n]) < 3 * floor["traj"] + TOL
        assert out["motion_pred"].shape[1] == 3 and tuple(out["motion_prob"].shape) == (out["motion_pred"].shape[0], 3)
    finally:
        model.close()


# ------------------------------------------------------------------ the N > 1 branch of bench.py on one GPU
def test_bench_two_ranks_on_one_gpu():
    """bench.py --gpus 2 under torch.distributed.run with both ranks on cuda:0 and the metric gather over gloo
    (PS_BENCH_BACKEND=gloo PS_BENCH_SAME_DEVICE=1): the driver's N > 1 command line, the sharding, the gather and the
    max-over-ranks timing -- gathered metrics equal those of one rank holding all the scenes."""
    env = dict(os.environ, PS_BENCH_BACKEND="gloo", PS_BENCH_SAME_DEVICE="1")
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--scenes-per-gpu", "2"]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", *common],
                        capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    lines = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["scenes_per_gpu"] == 2
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--scenes-per-gpu", "4"],
                        capture_output=True, text=True, timeout=1200, env=dict(os.environ), cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    for k, v in one["rollout_metrics"].items():      # scenes 0..3 either way (i % 2 sharding): identical sums up to order
        assert abs(two["rollout_metrics"][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, two["rollout_metrics"][k], v)
    assert abs(two["value"] - 4 * 128 * 80 / (two["ms_per_step"] * 1e-3)) < 1e-3 * two["value"]


def test_bench_eight_ranks_on_one_gpu_with_uneven_shards():
    """8-rank readiness without an 8-GPU node: bench.py --gpus 8 under torch.distributed.run, all eight ranks
    on cuda:0, FIVE scenes over the eight ranks (i % 8: ranks 0..4 hold one scene, ranks 5..7 none) -- communicator set-up, ranks
    without a scene inside every collective, uneven shards in the gather, the max-over-ranks timing.  The gathered metrics equal
    those of one rank holding the five scenes."""
    env = dict(os.environ, PS_BENCH_BACKEND="gloo", PS_BENCH_SAME_DEVICE="1")
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--inflight", "1"]
    r8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                         "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--total-scenes", "5", *common],
                        capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r8.returncode == 0, r8.stdout[-3000:] + r8.stderr[-3000:]
    lines = [l for l in r8.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    eight = json.loads(lines[0])
    assert eight["n_gpus"] == 8 and eight["config"]["scenes_total"] == 5 and eight["config"]["scenes_per_gpu"] == 1
    assert eight["rollout_metrics"]["scenes"] == 5 and eight["rollout_metrics"]["agents"] == 5 * 128
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *common, "--scenes-per-gpu", "5"],
                        capture_output=True, text=True, timeout=1200, env=dict(os.environ), cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    for k, v in one["rollout_metrics"].items():
        if k == "rollout_ade":   # (the mean over UPDATES of the per-update mean: eight single-scene updates against one five-scene update)
            continue
        assert abs(eight["rollout_metrics"][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, eight["rollout_metrics"][k], v)
    assert abs(eight["value"] - 5 * 128 * 80 / (eight["ms_per_step"] * 1e-3)) < 1e-3 * eight["value"]


# ------------------------------------------------------------------ (f4) learnable relative positional encoding
@pytest.mark.parametrize("shape", ["small", "split_s2s", "policy_only"])
def test_learnable_rel_pe_vs_oracle(shape):
    """*.ATTN.LEARNABLE_PE (layers/fourier_embedding.py:11-54): the edge-MLP kernel (k_pe_learn) behind every edge set of the
    scene encoder, the generator and the policy -- stage outputs and the closed loop against the fp64 oracle (the oracle is
    pinned to the reference by tests/golden/ref_standins_small_lpe_b2.npz, which test_hip_parity checks as well).
    'split_s2s': >= 2048 scene tokens, the s2s layers take the split launches with 128-column rows; 'policy_only': the
    flags are independent per part of the model."""
    from golden_cases import SPECS
    from prosim_amd.engine import Engine
    spec = SPECS["small_lpe"]
    if shape == "policy_only":
        spec = spec.replace(enc_learnable_pe=False, dec_learnable_pe=False)
    w = weights.init_weights(spec, 0)
    if shape == "split_s2s":
        scene = synth.make_scene(spec, 40, 1100, batch=2, seed=31, goal=True, ragged=True)
    else:
        scene = synth.make_scene(spec, 24, 90, batch=3, seed=30, goal=True, tags=True, ragged=True, replay=0.2)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
        o32 = orc.rollout(w, spec, scene, dtype=torch.float32)
    floor = float((o32["traj"].double() - o64["traj"]).abs().max())
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        if shape == "split_s2s":
            assert eng.num_agents + eng.num_map_tokens >= 2048
        eng.encode_scene()
        assert err(eng.get("scene_tokens"), o64["trace"]["scene_tokens"].numpy()) < TOL
        eng.generate_policy()
        pm = scene["prompt_mask"].astype(bool)
        assert err(eng.padded("policy_emd")[pm], o64["policy_emd"].numpy()[pm]) < 2 * TOL
        eng.rollout()
        pol = eng.policy_rows
        e0 = err(eng.get("motion_pred")[0][pol], o64["motion_pred"][:int(pol.sum())].numpy())
        assert e0 < TOL
        d = np.abs(eng.padded("traj") - o64["traj"].numpy())[pm].reshape(int(pm.sum()), -1).max(1)
        record(f"learnable_pe/{shape}", replan0_max=e0, fp32_floor=floor, **per_agent(d))
        assert d.max() < 3 * floor + TOL and np.median(d) < floor + TOL, (d.max(), floor)
    finally:
        eng.close()


def test_k_chain16_results_do_not_depend_on_the_tiling():
    """From 4 rows per workgroup up k_chain16 has ONE summation order (a row's 16-edge tiles by parity, merged (even, odd):
    two waves at 4 rows, one wave in two parity runs from 8 rows up), so latency mode and throughput mode return the same
    bits -- trajectories, predictions, fused features."""
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 150, 96, batch=4, seed=71, goal=True, tags=True, ragged=True)
    eng = Engine(spec, w)
    try:
        eng.set_chain_impl(2)
        outs = []
        for rows in (4, 8, 11, 16):
            eng.set_chain_rows(rows)
            eng.set_scene(scene)
            eng.rollout()
            outs.append((eng.get("traj"), eng.get("motion_pred"), eng.get("fused")))
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a, b)
    finally:
        eng.close()


# ------------------------------------------------------------------ binary (agent-pair) conditions
@pytest.mark.parametrize("shape", ["mixed", "pairs_only", "hub"])
def test_pair_conditions_vs_oracle(shape):
    """'v2v_tag' conditions (condition_encoders.py:148-150, condition_attns.py:141-166): edges s -> t and t -> s in the
    condition layers' graph, pooled with the unary keys that share an edge.  'hub': one prompt is the target of 40 pairs
    (two 32-edge tiles into one destination); masked rows and tag values outside the used list make no edge.  The oracle
    is pinned to the reference by tests/golden/ref_standins_small_v2v_b2.npz (test_hip_parity checks the engine on it)."""
    from golden_cases import SPECS
    from prosim_amd.engine import Engine
    spec = SPECS["small_v2v"]
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 48 if shape == "hub" else 20, 60, batch=2, seed=41, goal=(shape == "mixed"), tags=(shape == "mixed"),
                             drag=(shape == "mixed"), v2v=True, ragged=True)
    if shape == "hub":
        c = scene["cond"]["v2v_tag"]
        pol = np.nonzero(scene["prompt_mask"][0])[0]
        n = min(40, len(pol) - 1)
        c["input"][0, :n, 0] = 2.0                                   # Merging
        c["prompt_idx"][0, :n, 0] = pol[1:n + 1]
        c["prompt_idx"][0, :n, 1] = pol[0]
        c["mask"][0, :n] = True
        c["mask"][0, n:] = False
        c["input"][1, :3, 0] = 1.0                                   # ParallelDriving: not among the used tags -> no entry
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.encode_scene()
        eng.generate_policy()
        pm = scene["prompt_mask"].astype(bool)
        assert err(eng.padded("policy_emd")[pm], o64["policy_emd"].numpy()[pm]) < 2 * TOL
        # the conditions matter: without them the embeddings are elsewhere
        plain = dict(scene, cond={k: v for k, v in scene["cond"].items() if k != "v2v_tag"})
        eng.set_scene(plain)
        eng.encode_scene(); eng.generate_policy()
        assert err(eng.padded("policy_emd")[pm], o64["policy_emd"].numpy()[pm]) > 1e-2
        eng.set_scene(scene)
        eng.rollout()
        A = eng.num_agents
        assert err(eng.get("motion_pred")[0], o64["motion_pred"][:A].numpy()) < TOL
    finally:
        eng.close()
