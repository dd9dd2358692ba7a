"""One demo-cache scene end to end on the GPU, without trajdata: agent table + VectorMap protobuf + scene metadata ->
batch (formatting.py / vecmap.py) -> closed-loop rollout with log-replay agents -> world-frame trajectories and the
validation metric against the log.  The flow of the reference's demo notebook (text_prompt_inference.ipynb) minus the text.

usage: python tools/demo_scene_rollout.py [--cache DIR/trajdata_cache/waymo_train --scene scene_3] [--policy 12] [--ckpt model.ckpt]
Without --cache the committed sample (tests/golden: scene_1) is used; without --ckpt seeded random weights."""
import argparse, glob, lzma, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import formatting as fmt, vecmap as vm, weights
from prosim_amd.engine import Engine
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.distributed import reduce_pair_metrics, rows_to_slots

ap = argparse.ArgumentParser()
ap.add_argument("--cache"); ap.add_argument("--scene", default="scene_1"); ap.add_argument("--policy", type=int, default=12)
ap.add_argument("--t0", type=int, default=10); ap.add_argument("--ckpt")
a = ap.parse_args()
spec = DEMO_SPEC
gold = os.path.join(ROOT, "tests", "golden")
if a.cache:
    import pyarrow.ipc as ipc
    sd = os.path.join(a.cache, a.scene)
    with open(os.path.join(sd, "agent_data_dt0.10.feather"), "rb") as f:
        t = ipc.open_file(f).read_all()
    cols = {c: t.column(c).to_numpy(zero_copy_only=False) for c in t.column_names}
    origin = np.array([cols["x"][0], cols["y"][0]], np.float64)
    cols["x"], cols["y"] = cols["x"] - origin[0], cols["y"] - origin[1]
    with open(os.path.join(sd, "tls_data_dt0.10.feather"), "rb") as f:
        tl = ipc.open_file(f).read_all()
    tls = vm.tls_at(tl.column("lane_id").to_numpy(zero_copy_only=False), tl.column("scene_ts").to_numpy(zero_copy_only=False),
                    tl.column("status").to_numpy(zero_copy_only=False), a.t0)
    meta = fmt.agent_types_from_scene_metadata(os.path.join(sd, "scene_metadata_dt0.10.dill"))
    pb = open(os.path.join(a.cache, "maps", "waymo_train_" + a.scene.split("_")[1] + ".pb"), "rb").read()
else:
    g = np.load(os.path.join(gold, f"demo_{a.scene}_agent_table.npz"))
    cols, origin = {k: g[k] for k in g.files if k != "origin"}, g["origin"].astype(np.float64)
    tl = np.load(os.path.join(gold, f"demo_{a.scene}_tls_table.npz"))
    tls = vm.tls_at(tl["lane_id"], tl["scene_ts"], tl["status"], a.t0)
    meta = fmt.agent_types_from_scene_metadata(os.path.join(gold, f"demo_{a.scene}_metadata.dill"))
    mp_ = glob.glob(os.path.join(gold, "demo_waymo_train_" + a.scene.split("_")[1] + "_map.pb*"))[0]
    pb = open(mp_, "rb").read()
    pb = lzma.decompress(pb) if mp_.endswith(".xz") else pb
tr = fmt.tracks_from_table(cols)
t0 = a.t0
f = fmt.ego_frame(tr, t0)
pres = np.isfinite(tr["x"]) & np.isfinite(tr["heading"])
ever = [i for i in range(pres.shape[0]) if any(t0 + t < pres.shape[1] and pres[i, t0 + t] for t in spec.all_t_indices)]
ego = list(tr["agent_ids"]).index("ego")
stay = [i for i in ever if pres[i, t0:t0 + spec.max_steps + 1].all()]
policy = [ego] + [i for i in stay if i != ego][:a.policy - 1]
replay = [i for i in ever if i not in policy]
types = np.array([min(max(meta.get(str(x), 1), 1), 3) for x in tr["agent_ids"]], np.int64)
world = np.array([f[0] + origin[0], f[1] + origin[1], f[2]])
mp = vm.map_for_scene(spec, pb, world, tls=tls)
sc = fmt.rollout_batch_from_tracks(spec, tr, t0, policy, replay, frame=f, map_fields=mp, agent_types=types)
ids = sc.pop("agent_ids")
sc["cond"] = fmt.conditions_from_tracks(spec, tr, t0, policy + replay, sc["prompt_mask"][0])
w = weights.init_weights(spec, 0)
if a.ckpt:
    w = weights.from_state_dict(spec, torch.load(a.ckpt, map_location="cpu")["state_dict"])
eng = Engine(spec, w)
eng.set_scene(sc)
eng.rollout()
c, s = np.cos(f[2]), np.sin(f[2])
tf = np.array([[c, -s, world[0]], [s, c, world[1]], [0, 0, 1]], np.float32)       # scene-centre frame -> world
wt = eng.world_trajs(tf)
gt = fmt.pair_targets_from_tracks(spec, tr, t0, policy + replay)
slots, A, pol = eng.row_slots, eng.num_agents, eng.policy_rows
dev = torch.device("cuda", 0)
t_tgt = torch.from_numpy(np.ascontiguousarray(gt["tgt"][0][:, slots])).to(dev)
t_msk = torch.from_numpy(np.ascontiguousarray((gt["mask"][0][:, slots] & pol[None]).astype(np.uint8))).to(dev)
out = torch.zeros(A, 10, device=dev)
eng.pair_metric(out.data_ptr(), t_tgt.data_ptr(), t_msk.data_ptr())
eng.sync()
m = reduce_pair_metrics(rows_to_slots(out.cpu(), torch.from_numpy(slots), 1, len(slots)))
ms, st = eng.time_rollout(1, 5)
print(f"{a.scene}: {len(policy)} policy agents + {len(replay)} log-replay agents, {eng.num_map_tokens} map tokens; rollout {ms:.2f} ms "
      f"(encode {st[0]:.2f}, generate {st[1]:.2f}, replans {st[2]:.2f})")
print("metric vs the log (" + ("checkpoint" if a.ckpt else "random-init weights") + "):", {k: round(float(v), 3) for k, v in m.items()})
j = int(np.nonzero(pol)[0][0])
print(f"ego world path: start ({wt[j, 0, 0]:.1f}, {wt[j, 0, 1]:.1f}) -> end ({wt[j, -1, 0]:.1f}, {wt[j, -1, 1]:.1f}), logged end "
      f"({tr['x'][ego, t0 + spec.max_steps] + origin[0]:.1f}, {tr['y'][ego, t0 + spec.max_steps] + origin[1]:.1f})")
eng.close()
