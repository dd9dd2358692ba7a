"""What the per-step metric gather costs the pipelined rollouts (bench.py's N > 1 path), piece by piece: the same
4-engines-in-flight loop with (a) nothing, (b) the slot scatter, (c) the gather's copies, on the default stream and on a
side stream.  Prints ms per step and the host time spent inside the gather calls."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prosim_amd import synth, weights
from prosim_amd.engine import Engine
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.distributed import SceneMetricGather, rows_to_slots

spec, S, NFL = DEMO_SPEC, 8, 4
w = weights.init_weights(spec, 0)
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(S)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]})
         for k in parts[0]}
engines = [Engine(spec, w) for _ in range(NFL)]
for e in engines:
    e.set_chain_rows(16); e.set_scene(scene); e.rollout()
for e in engines:
    e.sync()
A, N = engines[0].num_agents, scene["prompt_mask"].shape[1]
streams = [torch.cuda.ExternalStream(e.stream_handle) for e in engines]
bufs = [torch.zeros(A, 10, device="cuda") for _ in range(NFL)]
slots = torch.from_numpy(engines[0].row_slots).cuda()
gather = SceneMetricGather(list(range(S)), S, N, 10, "cuda")
side = torch.cuda.Stream()

def run(mode, where, steps=24, wait_read=True):
    done = [torch.cuda.Event() for _ in range(NFL)]
    read = [None] * NFL
    host = 0.0
    def step(k):
        nonlocal host
        i = k % NFL
        if read[i] is not None and wait_read:
            streams[i].wait_event(read[i])
        engines[i].rollout()
        done[i].record(streams[i])
        if mode == "none":
            return
        t = time.perf_counter()
        st = {"default": torch.cuda.current_stream(), "side": side, "engine": streams[i]}[where]
        with torch.cuda.stream(st):
            if where != "engine":
                st.wait_event(done[i])
            if mode != "events":
                x = rows_to_slots(bufs[i], slots, S, N)
                if mode == "gather":
                    x = gather(x)
            if where != "engine":
                read[i] = torch.cuda.Event()
                read[i].record(st)
        host += time.perf_counter() - t
    for k in range(NFL):
        step(k)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:7s} on {where:8s} wait_read={int(wait_read)}: {1e3 * dt / steps:.3f} ms/step, host time in the gather calls {1e3 * host / steps:.3f} ms/step")

run("none", "side")
run("events", "side")
run("events", "side", wait_read=False)
run("slots", "side")
run("slots", "side", wait_read=False)
run("slots", "engine")
run("gather", "engine")
run("gather", "side")
run("none", "side")
