"""Are concurrent rollouts on several engines deterministic?  N engines hold the same resident batch; rollouts are issued round-robin
without synchronisation (bench.py's loop), optionally with torch work on the engines' streams and on a side stream; every
engine's motion_pred is digested after each round and compared with the first one."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
w = weights.init_weights(spec, 0)
S = 8
parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(S)]
scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
             {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
n = int(os.environ.get("PS_ENGINES", "4")); rounds = int(os.environ.get("PS_ROUNDS", "40")); mode = os.environ.get("PS_MODE", "plain")
engs = [Engine(spec, w) for _ in range(n)]
for e in engs:
    e.set_chain_rows(16); e.set_scene(scene); e.rollout()
for e in engs: e.sync()
want = hashlib.sha256(engs[0].get("motion_pred").tobytes()).hexdigest()[:12]
streams = [torch.cuda.ExternalStream(e.stream_handle) for e in engs]
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
if mode.startswith("nccl"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("nccl", rank=0, world_size=1)
    gbuf = torch.zeros(8, 128, 10, device="cuda"); gall = torch.zeros(8, 128, 10, device="cuda")
    gstream = torch.cuda.Stream()
bad = 0
for r in range(rounds):
    for rep in range(3):
        for i, e in enumerate(engs):
            if mode == "events" and rep == 1 and r % 4 == 1: e.enable_policy_events(True)
            if mode == "events" and rep == 2 and r % 4 == 1: e.enable_policy_events(False)
            e.rollout()
            if mode == "nccl" and rep == 1:        # the collective on the ENGINE's stream (bench.py until round 5)
                with torch.cuda.stream(streams[i]):
                    dist.all_gather_into_tensor(gall, gbuf)
            if mode == "nccl_side" and rep == 1:   # ... on a stream of its own that waits for the engine's
                gstream.wait_stream(streams[i])
                with torch.cuda.stream(gstream):
                    dist.all_gather_into_tensor(gall, gbuf)
            if mode in ("torch", "events"):
                with torch.cuda.stream(streams[i]):
                    t = torch.full((1024, 128), float("nan"), device="cuda"); t.index_copy_(0, torch.arange(512, device="cuda"), torch.zeros(512, 128, device="cuda"))
                with torch.cuda.stream(side):
                    junk2 = junk @ junk
    for e in engs: e.sync()
    if mode == "get2d":
        for e in engs: e.get("traj")
    if mode == "tcpu":
        _ = float(junk[0, 0].double().cpu()); _ = junk[:64].cpu()
    if mode == "pinned":
        pin = torch.empty(1 << 20, pin_memory=True); pin.copy_(junk.view(-1)[:1 << 20], non_blocking=True); torch.cuda.synchronize()
    got = [hashlib.sha256(e.get("motion_pred").tobytes()).hexdigest()[:12] for e in engs]
    if any(g != want for g in got):
        bad += 1
        print("round", r, "mismatch:", [g == want for g in got], flush=True)
print("mode %s engines %d rounds %d: rounds with a mismatching engine: %d (queues %s)" % (mode, n, rounds, bad, os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
