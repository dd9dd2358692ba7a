"""How close does the reference's own arithmetic come to its +-pi discontinuities on a scene?  (CPU; fp64 oracle.)
Every wrap_angle argument and every atan2 of the relative-PE inputs of one rollout is checked: the distance of the argument from
the cut (an odd multiple of pi for wrap_angle; pi - |angle| for the atan2 of angle_between_2d_vectors), smallest first, with the
call (edge set / replan) and the row it belongs to.  An fp32 evaluation carries ~1e-6 rad of noise in these arguments: a row listed
below ~1e-5 can land on either side of the cut in ANY fp32 implementation (the reference's included), and its features -- sines and
cosines of multiples of the wrapped angle -- then differ at order 1.
usage: python tools/cut_margin.py [config] [seed] [threshold]      (default: BASELINE configs[2], seed 5 = scene 5 of bench.py's batch)"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from oracle import prosim_oracle as orc

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
spec = DEMO_SPEC
if os.environ.get("NOTRUNC"):   # the no-truncation variant of the config (tests/test_round2_gpu.py): neighbour caps >= every candidate count
    cap_ = synth.BASELINE_CONFIGS[cfg]["n_agents"] + synth.BASELINE_CONFIGS[cfg]["n_polylines"]
    spec = DEMO_SPEC.replace(dec_max_neigh=cap_, pol_max_neigh=max(DEMO_SPEC.pol_max_neigh, min(cap_, 2047)))
w = weights.init_weights(spec, 0)
scene = synth.baseline_scene(spec, cfg, seed=seed, batch=1)
found, calls = [], {"wrap": 0, "pe": 0}
real_wrap, real_pe = orc.wrap_angle, orc.rel_pe_input


def wrap_rec(a):
    calls["wrap"] += 1
    if calls.get("in_pe"):
        return real_wrap(a)
    u = (a.detach().double() + math.pi) % (2 * math.pi)
    m = torch.minimum(u, 2 * math.pi - u)                       # distance of the argument from an odd multiple of pi
    idx = torch.nonzero(m < thr)
    for i in idx[:64]:
        found.append((float(m[tuple(i)]), f"wrap_angle call {calls['wrap']} shape {tuple(a.shape)}", tuple(int(x) for x in i)))
    return real_wrap(a)


def pe_rec(src, dst, ori_dst, pos_dst, ori_src, pos_src):
    calls["pe"] += 1
    calls["in_pe"] = True
    out = real_pe(src, dst, ori_dst, pos_dst, ori_src, pos_src)
    calls["in_pe"] = False
    ang = out[..., 2].detach().double()
    m = math.pi - ang.abs()
    for i in torch.nonzero(m < thr)[:64]:
        e = int(i[0])
        found.append((float(m[e]), f"atan2 of rel_pe_input call {calls['pe']} ({src.numel()} edges)", (int(dst[e]), int(src[e]))))
    u = ((ori_src[src] - ori_dst[dst]).detach().double().reshape(-1) + math.pi) % (2 * math.pi)   # the wrap_angle inside (rel_ori), with its edge
    m = torch.minimum(u, 2 * math.pi - u)
    for i in torch.nonzero(m < thr)[:64]:
        e = int(i[0])
        found.append((float(m[e]), f"rel_ori wrap of rel_pe_input call {calls['pe']} ({src.numel()} edges)", (int(dst[e]), int(src[e]))))
    return out


orc.wrap_angle, orc.rel_pe_input = wrap_rec, pe_rec
with torch.no_grad():
    orc.rollout(w, spec, scene, dtype=torch.float64)
orc.wrap_angle, orc.rel_pe_input = real_wrap, real_pe
found.sort(key=lambda t: t[0])
print(f"configs[{cfg}] seed {seed}: {calls['wrap']} wrap_angle calls, {calls['pe']} rel-PE input calls; arguments within {thr:g} rad of a cut: {len(found)}")
for m, where, idx in found[:40]:
    print(f"  {m:.3e} rad  {where}  (dst row, src row) / index {idx}")
