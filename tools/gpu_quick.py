"""Quick GPU check: attention layer parity (all T), config-2 rollout error vs fp32 oracle, timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC, DEMO_SPEC
from prosim_amd.engine import Engine
from oracle import prosim_oracle as orc
torch.set_num_threads(32)
spec = SMALL_SPEC
w = weights.init_weights(spec, 0); Wt = orc.W(w)
eng = Engine(spec, w)
g = torch.Generator().manual_seed(0)
def rnorm(r):
    return ((r - r.mean(-1, keepdim=True)) / torch.sqrt(r.var(-1, unbiased=False, keepdim=True) + 1e-5)).numpy()
for (grp, pre, bip) in (("s2s", "scene_encoder.s2s_attn_layers.0", False), ("a2p", "policy.act_decoder.a2p_attn_layers.1", True)):
    for (Ns, Nd, E) in ((300, 37, 2000), (50, 9, 0)):
        xs = torch.randn(Ns, 128, generator=g); xd = torch.randn(Nd, 128, generator=g); r = torch.randn(max(E, 1), 128, generator=g)[:E]
        src = torch.randint(0, Ns, (E,), generator=g); dst = torch.sort(torch.randint(0, max(Nd - 1, 1), (E,), generator=g))[0]
        ref = orc.attention_layer(Wt, pre, spec, xs, xd, r, src, dst, bip).numpy()
        eoff = np.zeros(Nd + 1, np.int64); np.add.at(eoff, dst.numpy() + 1, 1); eoff = np.cumsum(eoff)
        for T in (11, 2, 4, 16):   # rows per workgroup of k_attn_chain; 11 = 1 row on 4 waves, two workgroups per CU; 16 = split layer (k_node + k_edge_small) when degree <= 128
            out = eng.test_attn(eng.layer_index(grp, int(pre[-1])), xs.numpy(), xd.numpy(), rnorm(r) if E else np.zeros((1, 128), np.float32), eoff, src.numpy(), T)
            print(grp, Ns, Nd, E, "T", T, "err", np.abs(out - ref).max())
eng.close()
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
eng = Engine(spec, w)
for cfg, batch in ((2, None), (2, 8)):
    scene = synth.baseline_scene(spec, cfg, seed=0, batch=batch)
    eng.set_scene(scene)
    eng.rollout(); eng.sync()
    if batch is None:
        with torch.no_grad():
            o = orc.rollout(w, spec, scene)
        print("cfg", cfg, "traj err vs o32", np.abs(eng.padded("traj") - o["traj"].numpy()).max(),
              "mp", np.abs(eng.get("motion_pred").reshape(-1,1,10,5) - o["motion_pred"].numpy()).max())
    ms, st = eng.time_rollout(2, 10)
    A = eng.num_agents
    print(f"cfg {cfg} batch {batch}: {ms:.3f} ms/rollout stages {[round(x,3) for x in st]} chain {eng.time_policy_kernel(2):.4f} ms -> {A*80/ms*1e3:.3e} agent-steps/s", flush=True)
eng.close()
