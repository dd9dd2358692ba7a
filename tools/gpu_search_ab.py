"""A/B of the neighbour-search launch forms (ps_set_search_impl 0 = one launch per search, k_radius_geo; 1 = count / fill / record launches):
digests of the closed-loop results (must be equal), graph nodes and rollout times -- one configs[2] scene, the 8-scene bench batch in
latency and throughput mode, and a small ragged scene with conditions (p2p self matches, candidate filters)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from prosim_amd.engine import Engine

def dig(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]

def cat(parts):
    return {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
                {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}

bad = 0
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
one = synth.baseline_scene(spec, 2, seed=0, batch=1)
eight = cat([synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)])
eng = Engine(spec, w)
for name, scene, rows in (("1 scene", one, 0), ("8 scenes", eight, 0), ("8 scenes", eight, 16)):
    d = {}
    for impl in (1, 0, 1, 0):
        eng.set_search_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
        dd = (dig(eng.padded("traj")), dig(eng.get("motion_pred")))
        nodes = eng.graph_nodes
        ms, st = eng.time_rollout(3, 20)
        print("%-8s rows %2d search_impl %d: traj %s motion_pred %s | %3d graph nodes | rollout %.3f ms (enc %.3f gen %.3f loop %.3f)" % (
            name, rows, impl, dd[0], dd[1], nodes, ms, st[0], st[1], st[2]), flush=True)
        d.setdefault(impl, dd)
        bad += d[impl] != dd
    bad += d[0] != d[1]
eng.close()
spec = SMALL_SPEC
w = weights.init_weights(spec, 0)
scene = synth.make_scene(spec, 24, 160, batch=3, seed=5, goal=True, tags=True, ragged=True)
eng = Engine(spec, w)
for impl_c, rows in ((2, 0), (2, 16), (0, 0)):
    d = {}
    for impl in (1, 0):
        eng.set_search_impl(impl); eng.set_chain_impl(impl_c); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
        d[impl] = (dig(eng.padded("traj")), dig(eng.get("motion_pred")))
        print("small chain_impl %d rows %2d search_impl %d: traj %s motion_pred %s | %3d graph nodes" % (impl_c, rows, impl, d[impl][0], d[impl][1], eng.graph_nodes), flush=True)
    bad += d[0] != d[1]
eng.close()
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
