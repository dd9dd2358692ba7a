"""Randomised repeatability sweep: random scene shapes, options and MODEL VARIANTS (every variant set of tools/gpu_fuzz_parity.py: OBS_UPDATE forms, learnable
rel-PE, binary tags, K > 1 heads, noise + GMM, PRED_MODE cluster / mlp, PRED_VEL off, knn edges, small caps) -- each case rolled out ONCE with the GPU otherwise
idle, then three times beside three engines that replay configs[2] rollouts (operand-image chains: the load that exposed the packed-op_sel erratum in round 6), and
every result compared with the idle one BIT FOR BIT.  The engine must be a pure function of its inputs in every kernel it has, not only on the bench workload.
usage: python tools/gpu_fuzz_under_load.py [n_cases] [seed]      (FUZZ_ROWS=16: throughput mode on batches of >= 550 rows; FUZZ_IMPL / FUZZ_ROW_IMPL as the parity sweep)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import prosim_amd
prosim_amd.configure_runtime()
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC, DEMO_SPEC
from prosim_amd.engine import Engine

FUZZ_ROWS = int(os.environ.get("FUZZ_ROWS", "0"))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
S = SMALL_SPEC
variants = [S, S.replace(obs_fusion="mlp"), S.replace(obs_attn_update=True), S.replace(obs_fusion="mlp", obs_attn_update=True),
            S.replace(enc_learnable_pe=True, dec_learnable_pe=True, pol_learnable_pe=True), S.replace(pol_learnable_pe=True, obs_fusion="mlp"),
            S.replace(used_v2v_tags=("Following", "Merging", "Overtaking")), S.replace(motion_k=3, rollout_top_k=3),
            S.replace(pred_gmm=True, action_noise_std=0.05), S.replace(k_pred_mode="cluster", motion_k=3, rollout_top_k=3),
            S.replace(k_pred_mode="mlp", motion_k=2, rollout_top_k=2), S.replace(obs_attn_update=True, enc_learnable_pe=True),
            S.replace(pred_vel=False), S.replace(use_goal_pred_loss=False, obs_fusion="mlp"),
            S.replace(rel_pos_edge_func="knn"), S.replace(rel_pos_edge_func="knn", dec_max_neigh=12, pol_max_neigh=9),
            S.replace(goal_pred_k=4)]
wl = weights.init_weights(DEMO_SPEC, 0)
load = []
for k in range(3):
    e = Engine(DEMO_SPEC, wl); e.set_chain_impl(int(os.environ.get("PS_LOAD_IMPL", "1"))); e.set_scene(synth.baseline_scene(DEMO_SPEC, 2, seed=10 + k, batch=1)); e.rollout(); load.append(e)
NAMES = ("scene_tokens", "policy_emd", "motion_pred")
def results(eng):
    out = [eng.get(n).copy() for n in NAMES] + [eng.padded("traj").copy(), eng.padded("vel").copy()]
    if eng.spec.goal_pred_k > 0: out += [eng.get("goal_prob").copy(), eng.get("goal_point").copy()]
    return out
engines, bad, done = {}, [], 0
t0 = time.time()
for case in range(n_cases):
    spec = variants[rng.randint(len(variants))]
    if rng.rand() < 0.3 and spec.rel_pos_edge_func == "radius":
        spec = spec.replace(dec_max_neigh=int(rng.choice([4, 16, 512])), pol_max_neigh=int(rng.choice([3, 12, 768])), scene_knn=int(rng.choice([2, 8, 32])),
                            dec_prompt_radius=float(rng.choice([20.0, 300.0])), pol_agent_radius=float(rng.choice([15.0, 100.0])), pol_map_radius=float(rng.choice([10.0, 50.0])))
    kw = dict(n_agents=int(rng.choice([1, 3, 7, 16, 33, 64, 100, 150])), n_polylines=int(rng.choice([1, 5, 40, 128, 300, 600])), batch=int(rng.choice([1, 2, 3, 5])),
              seed=int(rng.randint(1 << 20)), goal=bool(rng.rand() < 0.5), tags=bool(rng.rand() < 0.4), drag=bool(rng.rand() < 0.4), ragged=bool(rng.rand() < 0.6),
              clustered=bool(rng.rand() < 0.5), replay=float(rng.choice([0.0, 0.0, 0.3, 0.6])), square=float(rng.choice([30.0, 100.0, 200.0])))
    if FUZZ_ROWS: kw["n_agents"], kw["batch"] = int(rng.choice([110, 130, 150])), int(rng.choice([5, 6, 8]))
    if kw["n_agents"] < 3: kw["replay"] = 0.0
    if kw["replay"] > 0 and rng.rand() < 0.5: kw["enter"] = 0.5
    if spec.used_v2v_tags: kw["v2v"] = True
    try:
        scene = synth.make_scene(spec, **kw)
    except Exception as ex:
        print(case, "skip (generator):", type(ex).__name__, ex, flush=True)
        continue
    if spec.motion_k > 1:
        scene["mode_choice"] = rng.randint(0, spec.motion_k, (spec.n_replans,) + scene["prompt_mask"].shape).astype(np.int32)
    if spec.action_noise_std > 0:
        scene["action_noise"] = (rng.standard_normal((spec.n_replans,) + scene["prompt_mask"].shape + (spec.motion_k, spec.target_steps, 2)) * spec.action_noise_std).astype(np.float32)
    key = repr(spec)
    if len(engines) > 10:
        for e in engines.values(): e.close()
        engines.clear()
    if key not in engines:
        engines[key] = Engine(spec, weights.init_weights(spec, 0))
        engines[key].set_row_impl(int(os.environ.get("FUZZ_ROW_IMPL", "0")))
        engines[key].set_chain_impl(int(os.environ.get("FUZZ_IMPL", "0")))
        engines[key].set_chain_rows(FUZZ_ROWS)
    eng = engines[key]
    for e in load: e.sync()
    eng.set_scene(scene); eng.rollout(); eng.sync()
    want = results(eng)
    diffs = []
    for rep in range(3):
        for e in load:
            for _ in range(3): e.rollout()
        eng.rollout()
        for nm, a, b in zip(NAMES + ("traj", "vel", "goal_prob", "goal_point"), results(eng), want):
            if not np.array_equal(a, b): diffs.append((rep, nm, float(np.abs(a - b).max())))
    done += 1
    if diffs:
        bad.append((case, key, kw, diffs))
        print(case, "DIFFERS under load:", diffs[:4], key[:120], kw, flush=True)
    elif case % 20 == 0:
        print(case, "ok", flush=True)
for e in list(engines.values()) + load: e.close()
print("repeatability under load: %d cases (x 3 runs beside 3 engines), %d differ from their idle run; %.0f s" % (done, len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
