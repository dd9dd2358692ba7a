"""Row-tile PointNet (ps_rowtile.h) against the reference-pure fixture and the staged kernel, by row tiles per wave; timings at
the benchmark's sizes (8192 map polylines x 19 points, 1024 agents x 11 steps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
eng = Engine(spec, w)
g = np.load(os.path.join(ROOT, "tests", "golden", "ref_pure_primitives.npz"))
for which, tag in ((0, "map"), (1, "obs")):
    x, m = g[f"pointnet_{tag}_x"], g[f"pointnet_{tag}_mask"]
    x = x.reshape(-1, *x.shape[2:]); m = m.reshape(-1, m.shape[2])
    valid = m.any(-1)
    ref = g[f"pointnet_{tag}_y"].reshape(-1, 128)
    for mt in (-1, 1, 2, 3, 4, 5, 0):
        if mt > 0 and 16 * mt < m.shape[1]: continue
        y, _ = eng.test_pointnet_mt(which, x, m, mt)
        print(f"{tag} P={m.shape[1]} n={m.shape[0]} mt={mt:2d}: max err vs reference {np.abs(y[valid] - ref[valid]).max():.2e}  invalid rows zero: {bool((y[~valid] == 0).all())}", flush=True)
rng = np.random.RandomState(0)
for which, n, P, C in ((0, 8192, 19, spec.map_dim), (1, 1024, 11, spec.obs_dim), (0, 1024, 19, spec.map_dim), (1, 128, 11, spec.obs_dim)):
    x = rng.randn(n, P, C).astype(np.float32)
    m = rng.rand(n, P) > 0.1
    base = None
    for mt in (-1, 1, 2, 3, 4, 5, 0):
        if mt > 0 and 16 * mt < P: continue
        y, ms = eng.test_pointnet_mt(which, x, m, mt, iters=20)
        if base is None: base = y
        print(f"which={which} n={n} P={P} mt={mt:2d}: {ms*1e3:8.1f} us   max diff vs staged {np.abs(y - base).max():.2e}", flush=True)
eng.close()
