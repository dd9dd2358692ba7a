"""How close does the reference's own arithmetic come to its +-pi discontinuities on a scene?  (CPU; fp64 oracle, oracle/cut_margin.py.)
Every relative-PE edge of one rollout is checked -- the wrap_angle of the heading difference and the atan2 of the bearing -- and the
ones within `thr` rad of the cut are listed, smallest margin first, with the call (edge set / replan) and (destination row, source row).
An fp32 evaluation carries 1e-7 - 1e-6 rad of noise in these arguments, more once closed-loop positions differ by 1e-5 m: a row listed
below ~1e-5 can land on the other side of the cut, and its features -- sines and cosines of multiples of the wrapped angle -- then
differ at order 1.
usage: python tools/cut_margin.py [config] [seed] [threshold]      (default: BASELINE configs[2], seed 5 = scene 5 of bench.py's batch;
NOTRUNC=1: the no-truncation variant of the config)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from oracle.cut_margin import near_cut_edges

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
spec = DEMO_SPEC
if os.environ.get("NOTRUNC"):   # the no-truncation variant of the config (tests/test_round2_gpu.py): neighbour caps >= every candidate count
    cap_ = synth.BASELINE_CONFIGS[cfg]["n_agents"] + synth.BASELINE_CONFIGS[cfg]["n_polylines"]
    spec = DEMO_SPEC.replace(dec_max_neigh=cap_, pol_max_neigh=max(DEMO_SPEC.pol_max_neigh, min(cap_, 2047)))
w = weights.init_weights(spec, 0)
scene = synth.baseline_scene(spec, cfg, seed=seed, batch=1)
found, calls = near_cut_edges(w, spec, scene, thr)
print(f"configs[{cfg}] seed {seed}: {calls['pe']} rel-PE input calls; edges within {thr:g} rad of a cut: {len(found)}")
for m, kind, call, d, s_, n in found[:40]:
    print(f"  {m:.3e} rad  {kind:7s} call {call:2d} ({n} edges)  dst row {d}  src row {s_}")
