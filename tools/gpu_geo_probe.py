"""geo_record as a probe beside real rollouts of other engines (experiments library: ps_test_geo_probe).  Prints, per lane quarter, how many
iterations returned other statistics / other inputs than the thread's first one.  usage: PS_LIB=... python tools/gpu_geo_probe.py [launches] [iters]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from prosim_amd import synth, weights, engine
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 50
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
lib = engine.load_library()
lib.ps_test_geo_probe.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
load = []
if not os.environ.get("PS_NOLOAD"):
    for k in range(3):
        e = Engine(spec, w); e.set_chain_impl(int(os.environ.get("PS_LOAD_IMPL", "1"))); e.set_scene(synth.baseline_scene(spec, 2, seed=10 + k, batch=1)); e.rollout(); load.append(e)
p = Engine(spec, w); p.set_scene(synth.baseline_scene(spec, 2, seed=3, batch=1)); p.encode_scene(); p.sync()
tot = np.zeros(8, np.uint64)
for rep in range(launches):
    for e in load:
        for _ in range(2): e.rollout()
    out = (C.c_uint64 * 8)()
    rc = lib.ps_test_geo_probe(p.h, int(os.environ.get("PS_PROBE_WG", "128")), iters, 1, out)
    if rc: raise RuntimeError(lib.ps_last_error().decode())
    tot += np.array(list(out), np.uint64)
print("geo_record probe, %d launches x %d iterations x 128 x 128 threads: statistics differ by lane quarter %s, inputs differ %s" % (launches, iters, tot[:4].tolist(), tot[4:].tolist()), flush=True)
for e in load + [p]: e.close()
