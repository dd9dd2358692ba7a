"""How evenly the edge phase's row queue (16 rows per workgroup, 8 waves, longest row first) spreads the policy sets' 16-edge tiles over a workgroup's
waves on the bench workload: makespan / mean wave load per (workgroup, set), for a row cost of c0 + tiles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
eng = Engine(spec, w)
eng.set_chain_rows(16); eng.set_scene(scene); eng.rollout(); eng.sync()
f = eng.lib.ps_test_get_edges
f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]; f.restype = ctypes.c_int64
for which, name in ((4, "a2p"), (5, "m2p"), (3, "s2p"), (2, "p2p"), (0, "a2a")):
    cap = 1 << 22
    esrc, edst = np.empty(cap, np.int32), np.empty(cap, np.int32)
    n = f(eng.h, which, esrc.ctypes.data, edst.ctypes.data, None, cap)
    if n <= 0:
        print(name, "no edges", n); continue
    nq = int(edst[:n].max()) + 1
    deg = np.bincount(edst[:n], minlength=(nq + 15) // 16 * 16)
    tiles = (deg + 15) // 16
    print(f"{name}: {n} edges over {nq} rows; degree min / median / mean / max {deg[:nq].min()} / {int(np.median(deg[:nq]))} / {deg[:nq].mean():.1f} / {deg[:nq].max()}; tiles per row mean {tiles[:nq].mean():.2f}")
    for c0 in (0.5, 1.0):
        ratios = []
        for g in range(len(deg) // 16):
            cost = sorted((c0 + t for t in tiles[16 * g:16 * g + 16] if t > 0), reverse=True)
            load = [0.0] * 8
            for c in cost:   # the queue: a wave that runs dry takes the next (longest remaining) row
                load[int(np.argmin(load))] += c
            if sum(load) > 0:
                ratios.append(max(load) / (sum(load) / 8))
        r = np.array(ratios)
        print(f"   row cost {c0} + tiles: makespan / mean wave load  mean {r.mean():.3f}  median {np.median(r):.3f}  p90 {np.percentile(r, 90):.3f}  max {r.max():.3f}  ({r.size} workgroups)")
eng.close()
