"""Per-replan error growth + edge-count comparison for one scene (diagnose closed-loop divergence)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
torch.set_num_threads(32)
cases = {
 "c29": (SMALL_SPEC, dict(n_agents=33, n_polylines=1, batch=5, seed=221310, goal=True, tags=True, drag=False, ragged=False, clustered=False, replay=0.0, square=200.0)),
 "c25": (SMALL_SPEC, dict(n_agents=100, n_polylines=300, batch=1, seed=809746, goal=False, tags=True, drag=True, ragged=False, clustered=False, replay=0.3, square=30.0)),
 "c50": (SMALL_SPEC.replace(obs_fusion="mlp"), dict(n_agents=150, n_polylines=40, batch=5, seed=377116, goal=False, tags=False, drag=True, ragged=False, clustered=False, replay=0.3, square=100.0)),
}
cases["t70"] = (SMALL_SPEC.replace(obs_fusion="mlp"), dict(n_agents=1, n_polylines=300, batch=5, seed=189543, goal=False, tags=False, drag=False, ragged=False, clustered=True, replay=0.0, square=30.0))
cases["t80"] = (SMALL_SPEC, dict(n_agents=3, n_polylines=128, batch=5, seed=587658, goal=False, tags=False, drag=False, ragged=True, clustered=False, replay=0.0, square=100.0))
# round 2, k_chain16 forced for every chain (PS_IMPL=2): the one case of 160 the sweep flagged
cases["r2_131"] = (SMALL_SPEC, dict(n_agents=33, n_polylines=5, batch=3, seed=95645, goal=False, tags=False, drag=False, ragged=True, clustered=False, replay=0.3, square=100.0))
for name in sys.argv[1:] or list(cases):
    spec, kw = cases[name]
    scene = synth.make_scene(spec, **kw)
    w = weights.init_weights(spec, 0)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
        o32 = orc.rollout(w, spec, scene)
    eng = Engine(spec, w)
    eng.set_chain_impl(int(os.environ.get("PS_IMPL", "0")))
    eng.set_scene(scene)
    eng.encode_scene(); eng.generate_policy(); eng.reset_rollout()
    pol = eng.policy_rows
    A = int(pol.sum())
    R = spec.n_replans
    print(name, "A", A, "rows", eng.num_agents)
    for t in range(R):
        eng.policy_step(t)
        ec = eng.get("edge_counts")
        mp = eng.get("motion_pred")[t][pol]
        ref = o64["motion_pred"][t * A:(t + 1) * A].numpy()
        r32 = o32["motion_pred"][t * A:(t + 1) * A].numpy()
        d = np.abs(mp - ref).reshape(A, -1).max(1)
        d32 = np.abs(r32 - ref).reshape(A, -1).max(1)
        se64, se32 = o64["step_edges"][t], o32["step_edges"][t]
        print(f"  replan {t}: gpu-vs-o64 max {d.max():.2e} (agent {int(d.argmax())}, #>1e-4: {int((d > 1e-4).sum())}) | o32-vs-o64 max {d32.max():.2e} (#>1e-4: {int((d32 > 1e-4).sum())}) | edges gpu a2p {int(ec[4])} m2p {int(ec[5])} o64 {se64} o32 {se32}")
    eng.close()
