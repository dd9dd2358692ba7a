"""Which LAUNCH of the encoder stops repeating its bits under load?  Experiments build only (PS_LIB=prosim_amd/libprosim_hip_exp.so:
PS_DBG_STOP ends ps_encode_scene after its k-th stage, the dbg_* names of ps_get read the buffers the stage left).  Three engines replay full
rollouts as load (PS_LOAD_IMPL), or PS_POISON=1: one engine launches the poison kernel (LDS + registers filled with a bit pattern) instead;
three probes run the encoder up to stage k again and again and compare the stage's buffers with their first run, bit for bit.
usage: PS_LIB=... python tools/gpu_stage_bisect.py [iterations] [stages, e.g. 1,2,3,4,5]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from prosim_amd import synth, weights, engine
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
stages = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,5").split(",")]
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
lib = engine.load_library()
poison = int(os.environ.get("PS_POISON", "0"))
load = []
if not os.environ.get("PS_NOLOAD"):
    for k in range(1 if poison else 3):
        e = Engine(spec, w); e.set_chain_impl(int(os.environ.get("PS_LOAD_IMPL", "0"))); e.set_scene(synth.baseline_scene(spec, 2, seed=10 + k, batch=1)); e.rollout(); load.append(e)
if poison:
    lib.ps_test_poison.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_int32]
    pat = int(os.environ.get("PS_POISON_PAT", "0x7fff7fff"), 16)
probe = []
for k in range(3):
    e = Engine(spec, w)
    e.set_chain_impl(int(os.environ.get("PS_IMPL", "0")))
    e.set_scene(synth.baseline_scene(spec, 2, seed=3 + k, batch=1))
    probe.append(e)
# what each stage leaves (stage 3 + 3 i: k | v of a2a layer i; 4 + 3 i: the a2a layer; 5 + 3 i: the s2s layer)
def names(stage):
    if stage == 1: return ["scene_tokens"]
    if stage == 2: return ["dbg_esrc_a2a", "dbg_esrc_s2s", "dbg_geo_a2a"] + (["dbg_geo_s2s"] if int(os.environ.get("PS_IMPL", "0")) != 1 else [])
    if stage >= 3 and (stage - 3) % 3 == 0: return ["dbg_kv_agents", "dbg_kh_agents"]
    return ["scene_tokens"]
def raw(e, name):
    if name == "scene_tokens": return e.get(name)
    cap = 1 << 20
    buf = np.empty(cap, np.float32)
    cnt = lib.ps_get(e.h, name.encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), cap)
    if cnt < 0: raise RuntimeError(lib.ps_last_error().decode())
    return buf[:cnt].view(np.uint32).copy()
def run_stage(e, stage):
    os.environ["PS_DBG_STOP"] = str(stage)
    e.encode_scene()
    return [raw(e, nm) for nm in names(stage)]
def kick():
    if poison:
        rc = lib.ps_test_poison(load[0].h, 512, pat, 8)
        if rc: raise RuntimeError(lib.ps_last_error().decode())
    else:
        for e in load: e.rollout()
for stage in stages:
    ref = [run_stage(e, stage) for e in probe]
    bad = {nm: 0 for nm in names(stage)}
    nan = 0
    shown = 0
    for it in range(n):
        if load: kick()
        for k, e in enumerate(probe):
            out = run_stage(e, stage)
            for nm, o, r in zip(names(stage), out, ref[k]):
                if not np.array_equal(o.view(np.uint32), r.view(np.uint32)):
                    bad[nm] += 1
                    if o.dtype == np.float32 and not np.isfinite(o).all(): nan += 1
                    if shown < 6:
                        shown += 1
                        d = (o.view(np.uint32) != r.view(np.uint32)).reshape(-1)
                        idx = np.nonzero(d)[0]
                        print("  stage %d %s it %d engine %d: %d words differ, first at %s" % (stage, nm, it, k, len(idx), idx[:8].tolist()), flush=True)
    print("stage %d: of %d runs differ: %s  (non-finite: %d)" % (stage, 3 * n, bad, nan), flush=True)
os.environ["PS_DBG_STOP"] = "0"
for e in load + probe: e.close()
