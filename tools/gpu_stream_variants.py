"""Which piece of serving NEW batches costs throughput: the resident no-sync loop of bench.py, then + result copies, + events,
+ set_scene, at PS_DEPTH engines."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine

spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
S = 8; A = S * 128
batches = [synth.baseline_scene(spec, 2, seed=1000 + i, batch=S) for i in range(6)]
d = int(os.environ.get("PS_DEPTH", "4")); nb = 80; Q = 2
engs = [Engine(spec, w) for _ in range(d)]
for i, e in enumerate(engs):
    e.set_chain_rows(16); e.set_scene(batches[i % 6]); e.rollout()
for e in engs: e.sync()
streams = [torch.cuda.ExternalStream(e.stream_handle) for e in engs]
shp = {n: engs[0].result_shape(n) for n in ("traj", "vel")}
bufs = [[{n: torch.empty(int(np.prod(s)), dtype=torch.float32, pin_memory=True) for n, s in shp.items()} for _ in range(Q)] for _ in engs]
def run(copies, events, setscene, sync_before):
    evs = []
    t0 = time.perf_counter()
    for i in range(nb):
        k = i % d; e = engs[k]
        if events and len(evs) >= d * Q:
            evs.pop(0).synchronize()
        if sync_before: e.sync()
        if setscene: e.set_scene(batches[i % 6])
        e.rollout()
        if copies:
            b = bufs[k][(i // d) % Q]
            for n in shp: e.get_async(n, b[n].data_ptr(), b[n].numel())
        if events:
            ev = torch.cuda.Event(); ev.record(streams[k]); evs.append(ev)
    for e in engs: e.sync()
    return nb * A * spec.max_steps / (time.perf_counter() - t0) / 1e6
for name, args in (("resident, no sync (bench.py's loop)", (0, 0, 0, 0)), ("+ result copies (get_async)", (1, 0, 0, 0)),
                   ("+ an event per batch, waited 2 x depth behind", (1, 1, 0, 0)), ("+ set_scene per batch (the pipeline)", (1, 1, 1, 0)),
                   ("set_scene only", (0, 0, 1, 0)), ("resident, engine synchronised before reuse", (0, 0, 0, 1))):
    vals = [run(*args) for _ in range(3)]
    print("depth %d  %-48s %s M agent-steps/s" % (d, name, " ".join("%.2f" % v for v in vals)), flush=True)
