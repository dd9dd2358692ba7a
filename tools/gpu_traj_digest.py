"""SHA-256 digests of the closed-loop results of the bench workload (8 x configs[2], seeds 0..7) and of a small ragged scene
with conditions, by engine mode -- for bit-exact A/B between two builds of the library (PS_LIB=<other .so>)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from prosim_amd.engine import Engine

def dig(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, w)
for impl, rows in ((0, 0), (0, 16), (2, 1), (2, 2), (2, 8)):
    eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
    print("bench8 impl %d rows %2d traj %s motion_pred %s  policy launch %.4f ms" % (
        impl, rows, dig(eng.padded("traj")), dig(eng.get("motion_pred")), eng.time_policy_kernel(3)), flush=True)
eng.close()
spec = SMALL_SPEC
w = weights.init_weights(spec, 0)
scene = synth.make_scene(spec, 24, 160, batch=3, seed=5, goal=True, tags=True, ragged=True)
eng = Engine(spec, w)
for impl, rows in ((2, 0), (2, 4), (2, 16)):
    eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
    print("small impl %d rows %2d traj %s motion_pred %s" % (impl, rows, dig(eng.padded("traj")), dig(eng.get("motion_pred"))), flush=True)
eng.close()
