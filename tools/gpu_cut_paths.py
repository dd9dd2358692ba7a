"""One BASELINE configs[2] scene against the REFERENCE-made fixture of that scene (tests/golden/ref_standins_demo_cfg2_*.npz), by engine
path: which fused-chain kernel / rows per workgroup lands how many agents outside the 1e-4 band (near-cut edges, tools/cut_margin.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from prosim_amd import synth, weights
from prosim_amd.engine import Engine
from gen_golden import FULL_CASES, SPECS, GOLD
for name in sys.argv[1:] or ["demo_cfg2_b1", "demo_cfg2_seed5"]:
    if name in FULL_CASES:
        sname, kw, wseed = FULL_CASES[name]
    else:   # demo_cfg2_seed<N>: further scenes of the benchmark batch (fixtures from `tests/gen_golden.py cfg2_scenes N ...`, not committed)
        sname, kw, wseed = "demo", dict(n_agents=128, n_polylines=1024, batch=1, seed=int(name.rsplit("seed", 1)[1]), goal=True), 0
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    w = weights.init_weights(spec, wseed)
    scene = synth.make_scene(spec, **kw)
    eng = Engine(spec, w)
    for impl, rows in ((0, 0), (1, 0), (2, 1), (2, 2), (2, 4), (2, 8), (2, 16), (3, 16)):
        eng.set_chain_impl(impl); eng.set_chain_rows(rows); eng.set_scene(scene); eng.rollout(); eng.sync()
        d = np.abs(eng.padded("traj") - g["traj"]).max(axis=(2, 3))[0]
        out = np.nonzero(d >= 1e-4)[0]
        print(f"{name} impl {impl} rows {rows:2d}: max {d.max():.2e} median {np.median(d):.2e} outside 1e-4: {len(out)} {out.tolist()}", flush=True)
    eng.close()
