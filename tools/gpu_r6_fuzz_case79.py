"""Round 6: the one BAD case of the default parity sweep (seed 61, case 79: 3 scenes x 7 agents, ATTN_UPDATE + FUSION mlp, log-replay agents that enter): per-agent closed-loop
error of every engine path against the fp64 oracle, first replan at which an agent leaves the 1e-4 band, and the fp64 rollout's near-cut edges (oracle/cut_margin.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
from oracle.cut_margin import near_cut_edges
spec = SMALL_SPEC.replace(obs_fusion="mlp", obs_attn_update=True)
kw = {'n_agents': 7, 'n_polylines': 300, 'batch': 3, 'seed': 59750, 'goal': True, 'tags': True, 'drag': False, 'ragged': False, 'clustered': True, 'replay': 0.3, 'square': 200.0, 'enter': 0.5}
scene = synth.make_scene(spec, **kw)
w = weights.init_weights(spec, 0)
torch.set_num_threads(16)
with torch.no_grad():
    o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
    o32 = orc.rollout(w, spec, scene)
pm = scene["prompt_mask"].astype(bool)
ref = o64["traj"].numpy()
d32 = np.abs(o32["traj"].numpy() - ref)[pm]
print("policy agents (scene, slot):", [tuple(x) for x in np.argwhere(pm)])
print("fp32 oracle: per-agent max", np.round(d32.reshape(d32.shape[0], -1).max(1), 6).tolist())
edges, _ = near_cut_edges(w, spec, scene, 1e-4)
print("fp64 near-cut edges (margin rad, kind, replan?, ...):", sorted(edges)[:6])
for impl in (0, 1, 2):
    eng = Engine(spec, w); eng.set_chain_impl(impl); eng.set_scene(scene); eng.rollout()
    d = np.abs(eng.padded("traj") - ref)[pm]                      # [A, steps, 4]
    per_step = d.max(-1)                                          # [A, steps]
    first = [(int(np.argmax(r >= 1e-4)) if (r >= 1e-4).any() else -1) for r in per_step]
    print("impl %d: per-agent max %s | first step outside 1e-4 %s" % (impl, np.round(per_step.max(1), 6).tolist(), first))
    eng.close()
