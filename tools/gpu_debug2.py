import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
spec = SMALL_SPEC
w = weights.init_weights(spec, 0); Wt = orc.W(w)
eng = Engine(spec, w)
torch.manual_seed(0)
def rnorm(r):
    return ((r - r.mean(-1, keepdim=True)) / torch.sqrt(r.var(-1, unbiased=False, keepdim=True) + 1e-5)).numpy()
for (grp, pre, bip) in (("s2s", "scene_encoder.s2s_attn_layers.0", False), ("a2p", "policy.act_decoder.a2p_attn_layers.1", True)):
    for (Ns, Nd, E) in ((20, 7, 60), (300, 37, 2000), (50, 9, 0)):
        xs = torch.randn(Ns, 128); xd = torch.randn(Nd, 128); r = torch.randn(max(E, 1), 128)[:E]
        src = torch.randint(0, Ns, (E,)); dst = torch.sort(torch.randint(0, max(Nd - 1, 1), (E,)))[0]
        ref = orc.attention_layer(Wt, pre, spec, xs, xd, r, src, dst, bip).numpy()
        eoff = np.zeros(Nd + 1, np.int32); np.add.at(eoff, dst.numpy() + 1, 1); eoff = np.cumsum(eoff).astype(np.int32)
        for T in (1, 2, 4):
            out = eng.test_attn(eng.layer_index(grp, int(pre[-1])), xs.numpy(), xd.numpy(), rnorm(r) if E else np.zeros((1, 128), np.float32), eoff, src.numpy(), T)
            d = np.abs(out - ref)
            print(grp, Ns, Nd, E, "T", T, "err", d.max(), "rows", np.abs(out - ref).max(1)[:8].round(5))
# edge sets + rel-PE of the scene encoder
scene = synth.make_scene(spec, 16, 128, batch=2, seed=0, goal=True, tags=True, ragged=True)
eng.set_scene(scene); eng.encode_scene()
o = orc.rollout(w, spec, scene, collect=True)
import prosim_amd
tt = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(dt)
map_mask = tt(scene["map_mask"], torch.bool).any(-1); obs_mask = tt(scene["prompt_mask"], torch.bool)
m_pos = tt(scene["map_pos"])[map_mask]; o_pos = tt(scene["obs_pos"])[obs_mask]
m_ori = tt(scene["map_head"])[map_mask][:, None]; o_ori = tt(scene["obs_head"])[obs_mask][:, None]
mb = orc._flat_batch_idx(map_mask); ob = orc._flat_batch_idx(obs_mask)
s_pos = torch.cat([m_pos, o_pos]); s_ori = torch.cat([m_ori, o_ori]); sb = torch.cat([mb, ob])
d, s = orc.knn_edges(s_pos, sb, s_pos, sb, spec.scene_knn)
es, ed, rt = eng.get_edges(1)
ref_set = set(zip(d.tolist(), s.tolist())); got = set(zip(ed.tolist(), es.tolist()))
print("s2s edges", len(ref_set), len(got), "sym diff", len(ref_set ^ got))
pe = orc.rel_pe(spec, torch.from_numpy(es).long(), torch.from_numpy(ed).long(), s_ori, s_pos, s_ori, s_pos)
print("s2s rt err", np.abs(rnorm(pe) - rt).max())
Mv = int(map_mask.sum())
d, s = orc.knn_edges(o_pos, ob, o_pos, ob, spec.agent_knn)
es, ed, rt = eng.get_edges(0)
ref_set = set(zip(d.tolist(), (s + Mv).tolist())); got = set(zip(ed.tolist(), es.tolist()))
print("a2a edges", len(ref_set), len(got), "sym diff", len(ref_set ^ got))
pe = orc.rel_pe(spec, torch.from_numpy(es).long() - Mv, torch.from_numpy(ed).long(), o_ori, o_pos, o_ori, o_pos)
print("a2a rt err", np.abs(rnorm(pe) - rt).max())
