import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
spec = SMALL_SPEC.replace(dec_max_neigh=8, pol_max_neigh=5)
w = weights.init_weights(spec, 0)
scene = synth.make_scene(spec, 20, 64, batch=2, seed=9, square=60.0)
eng = Engine(spec, w); eng.set_scene(scene)
o = orc.rollout(w, spec, scene, collect=True)
eng.encode_scene()
print("scene_tokens", np.abs(eng.get("scene_tokens") - o["trace"]["scene_tokens"].numpy()).max())
eng.generate_policy()
print("policy_emd rows", np.abs(eng.get("policy_emd") - o["policy_emd"].reshape(-1,128).numpy()).max(1).round(4))
tt = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(dt)
mm, om = tt(scene["map_mask"], torch.bool).any(-1), tt(scene["prompt_mask"], torch.bool)
m_pos, o_pos = tt(scene["map_pos"])[mm], tt(scene["obs_pos"])[om]
mb, ob = orc._flat_batch_idx(mm), orc._flat_batch_idx(om)
s_pos, sb = torch.cat([m_pos, o_pos]), torch.cat([mb, ob]); Mv = int(mm.sum())
d, s = orc.radius_edges(o_pos, ob, o_pos, ob, spec.dec_prompt_radius, spec.dec_max_neigh, drop_self=True)
es, ed, _ = eng.get_edges(2)
ref = list(zip(d.tolist(), (s + Mv).tolist())); got = list(zip(ed.tolist(), es.tolist()))
print("p2p equal", ref == got, len(ref), len(got)); 
if ref != got:
    for q in range(40):
        r_ = [x[1]-Mv for x in ref if x[0]==q]; g_ = [x[1]-Mv for x in got if x[0]==q]
        if r_ != g_: print(q, r_, g_)
d, s = orc.radius_edges(s_pos, sb, o_pos, ob, spec.dec_scene_radius, spec.dec_max_neigh)
es, ed, _ = eng.get_edges(3)
print("s2p equal", list(zip(d.tolist(), s.tolist())) == list(zip(ed.tolist(), es.tolist())))
