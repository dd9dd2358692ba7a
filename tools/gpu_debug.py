"""Stage-by-stage comparison of the HIP engine against the oracle (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC, DEMO_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc

def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())

def run(spec, kw, wseed, tag):
    print(f"==== {tag}", flush=True)
    w = weights.init_weights(spec, wseed)
    scene = synth.make_scene(spec, **kw)
    with torch.no_grad():
        o = orc.rollout(w, spec, scene, collect=True)
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
    eng = Engine(spec, w)
    eng.set_scene(scene)
    pm = scene["prompt_mask"].astype(bool)
    eng.encode_scene()
    print("scene_tokens err", err(eng.get("scene_tokens"), o["trace"]["scene_tokens"].numpy()),
          "absmax", float(o["trace"]["scene_tokens"].abs().max()))
    eng.generate_policy()
    print("policy_emd err", err(eng.get("policy_emd"), o["policy_emd"][torch.from_numpy(pm)].numpy()),
          "absmax", float(o["policy_emd"].abs().max()))
    print("reconst err", err(eng.get("reconst_pred"), o["reconst_pred"].numpy()))
    ec = eng.get("edge_counts"); print("edges", ec, o["edges"])
    eng.reset_rollout()
    A = eng.num_agents
    for t in range(spec.n_replans):
        eng.policy_step(t)
        f = eng.get("fused"); mp = eng.get("motion_pred")[t]
        print(f" step {t}: fused err {err(f, o['trace']['fused'][t].numpy()):.3e}  motion_pred err "
              f"{err(mp, o['motion_pred'][t*A:(t+1)*A].numpy()):.3e} (vs f64 {err(mp, o64['motion_pred'][t*A:(t+1)*A].numpy()):.3e}; "
              f"oracle32 vs f64 {err(o['motion_pred'][t*A:(t+1)*A].numpy(), o64['motion_pred'][t*A:(t+1)*A].numpy()):.3e})"
              f" edges {eng.get('edge_counts')[4:6]} {o['step_edges'][t]}", flush=True)
    tr, vl = eng.padded("traj"), eng.padded("vel")
    print("traj err vs o32", err(tr, o["traj"].numpy()), "vs o64", err(tr, o64["traj"].numpy()),
          "| o32 vs o64", err(o["traj"].numpy(), o64["traj"].numpy()))
    print("vel  err vs o32", err(vl, o["vel"].numpy()), "vs o64", err(vl, o64["vel"].numpy()))
    # full rollout in one call + timing
    eng.rollout(); eng.sync()
    print("rollout() traj err vs o64", err(eng.padded("traj"), o64["traj"].numpy()))
    ms, st = eng.time_rollout(2, 5)
    print(f"time: {ms:.3f} ms/rollout stages {st}; policy chain kernel {eng.time_policy_kernel(2):.4f} ms", flush=True)
    eng.close()

if __name__ == "__main__":
    g = np.load(os.path.join(ROOT, "tests/golden/ref_pure_primitives.npz"))
    eng = Engine(DEMO_SPEC, weights.init_weights(DEMO_SPEC, 0))
    for which, tag in ((0, "map"), (1, "obs")):
        x, m = g[f"pointnet_{tag}_x"], g[f"pointnet_{tag}_mask"]
        y = eng.test_pointnet(which, x.reshape(-1, *x.shape[2:]), m.reshape(-1, m.shape[2]))
        ref = g[f"pointnet_{tag}_y"].reshape(-1, 128)
        valid = m.reshape(-1, m.shape[2]).any(-1)
        print("pointnet", tag, "err", err(y[valid], ref[valid]))
    print("fourier err", err(eng.test_fourier(g["fourier_x"]), g["fourier_y"]))
    print("wrap err", err(eng.test_wrap(g["wrap_x"]), g["wrap_y"]))
    eng.close()
    run(SMALL_SPEC, dict(n_agents=16, n_polylines=128, batch=2, seed=0, goal=True, tags=True, ragged=True), 0, "small ragged b2")
    run(SMALL_SPEC, dict(n_agents=16, n_polylines=128, batch=1, seed=1), 1, "small plain")
    run(DEMO_SPEC, dict(n_agents=64, n_polylines=512, batch=1, seed=0), 0, "demo 64/512")
    run(DEMO_SPEC, dict(n_agents=128, n_polylines=1024, batch=1, seed=0, goal=True), 0, "demo 128/1024 goal")
