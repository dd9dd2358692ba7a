"""Scene-token and policy-token error of the engine against the fp64 oracle on one configs[2] scene, by scene-encoder path (the split
s2s kernels, the fused chain, k_chain16): how closely each tracks the reference BEFORE the closed loop amplifies anything."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
spec = DEMO_SPEC
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w = weights.init_weights(spec, 0)
scene = synth.baseline_scene(spec, 2, seed=seed, batch=1)
torch.set_num_threads(32)
with torch.no_grad():
    o = orc.rollout(w, spec, scene, dtype=torch.float64, collect=True)
    o32 = orc.rollout(w, spec, scene, dtype=torch.float32, collect=True)
Mv = int(scene["map_mask"].any(-1).sum())   # map tokens: the rows of scene_tokens that no replan rewrites
ref_tok, ref_emd = o["trace"]["scene_tokens"].numpy()[:Mv], o["policy_emd"].numpy()
def stats(a, b):
    d = np.abs(a.astype(np.float64) - b)
    return f"max {d.max():.2e} rms {np.sqrt((d * d).mean()):.2e}"
print("fp32 oracle                        : map tokens", stats(o32["trace"]["scene_tokens"].numpy()[:Mv], ref_tok), "| policy_emd", stats(o32["policy_emd"].numpy(), ref_emd))
eng = Engine(spec, w)
pm = scene["prompt_mask"].astype(bool)
ref_traj = o["traj"].numpy()
for label, impl, rimpl in (("engine, staged row kernels (r3)", 0, 1), ("engine, row-tile kernels", 0, 0), ("engine, row-tile, node mt 1", 0, 11), ("engine, impl 3 (s2s on k_chain16)", 3, 1)):
    eng.set_row_impl(rimpl); eng.set_chain_impl(impl); eng.set_chain_rows(16 if impl == 3 else 0); eng.set_scene(scene); eng.rollout(); eng.sync()
    tok = eng.get("scene_tokens")[:Mv]
    emd = eng.padded("policy_emd")[pm]
    d = np.abs(eng.padded("traj").astype(np.float64) - ref_traj)[pm].reshape(int(pm.sum()), -1).max(1)
    print(f"{label:34s}: map tokens", stats(tok, ref_tok), "| policy_emd", stats(emd, ref_emd[pm]),
          f"| closed loop: {int((d >= 1e-4).sum())} agents outside 1e-4, max {d.max():.2e}, median {np.median(d):.2e}", flush=True)
eng.close()
