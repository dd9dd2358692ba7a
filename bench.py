#!/usr/bin/env python
"""bench.py -- closed-loop rollout throughput on MI355X (contract: see the task statement).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scenes-per-gpu S] [--config I]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full closed-loop rollout (scene encode + policy generator + 8 replans x 10 steps)
of S synthetic 128-agent / 1024-polyline scenes per GPU, inputs resident in HBM.  Default S = 8 is
BASELINE.json configs[3]'s per-GPU share (64 scenes sharded 8 per GPU; at --gpus 8 the job IS configs[3]);
every scene is a configs[2] scene (goal-point prompts).  metric = agent-steps/s over all ranks; the
single-scene (S = 1, latency-bound) figure is measured in the same run and reported alongside.
Scenes shard by index over ranks (i % world, rollout/callbacks.py:76) with no data-path collective;
after each rollout the per-agent sums of the reference's rollout metric (PairMotionPred, computed on the device against
seeded synthetic targets) are all-gathered over RCCL (the path's only exchange).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes its streams onto 4 hardware queues by default; the pipelined steps use one stream per engine
# plus torch's, and with 4 queues the engines' graph replays serialise behind one another (4 engines in flight: 9.5 M
# agent-steps/s with 4 queues, 16.7 M with 8).  Round 5: 16 -- every stream beyond the queue count SHARES a queue, and a job with
# RCCL's streams beside the engines', the upload stream and torch's own crosses 8 (the forced-distributed line fell from 26.8 to
# 15.4 M at 8 queues; prosim_amd.configure_runtime() does the same for other users of the package).  Round 6: 24, for 16 engines in flight
# (with 16 queues that many engines fell back to 27.7 M; with 24: 30.0 M).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from prosim_amd import synth, weights  # noqa: E402
from prosim_amd.postprocess import replicate_scene  # noqa: E402
from prosim_amd.spec import DEMO_SPEC  # noqa: E402

D = 128


def algorithmic_flops_chain(A: int, e_a2p: float, e_m2p: float, layers: int) -> float:
    """Reference-formulation FLOPs of the policy attention chain per launch:
    per AttentionLayer D^2 (26 N_d + 4 E) + 4 E D  (q 2, gate 4, s 2, out 2, FFN 16 per destination;
    per-edge to_k_r / to_v_r 4 D^2; scores + weighted sum 4 D).  The source k/v projections
    (4 N_s D^2) run in a separate kernel and are not counted here."""
    per = lambda E: D * D * (26 * A + 4 * E) + 4 * E * D
    return layers * (per(e_a2p) + per(e_m2p))


def algorithmic_bytes_chain(A: int, n_src_a2p: int, n_src_m2p: int, e_a2p: float, e_m2p: float, layers: int) -> dict:
    """Bytes one policy launch HAS to move: every layer's weights once (what the kernel streams per layer: the pre-split
    fp16 hi | lo fragments + small vectors, 960 KB), the k | v rows of every source token once per layer (k as split fp16 512 B + v as fp32
    512 B), the 32-byte geometry record of every edge once (the six layers of a set re-read the same records: a workgroup's share stays in
    its XCD's L2), and the destination rows in and out.  Gathers that revisit a source row (62 a2p / 160 m2p edges per destination) are
    served by L2 and do not count either."""
    w = 2 * layers * 960 * 1024
    kv_a = layers * n_src_a2p * 1024
    kv_m = layers * n_src_m2p * 1024
    rec = int(e_a2p + e_m2p) * 32
    xio = 2 * A * 128 * 4
    return {"weights": w, "kv_rows_a2p": kv_a, "kv_rows_m2p": kv_m, "geometry_records": rec, "rows_in_out": xio, "total": w + kv_a + kv_m + rec + xio}


def newest_pmc_json():
    """profiles/r??_pmc_policy_chain.json of the latest round (tools/make_pmc_json.py): the offline counter passes behind roofline.traffic."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_policy_chain.json")))
    return files[-1] if files else None


def pin_to_gpu_numa_node(dev_index: int):
    """N > 1 ranks on one node: keep this rank's host threads (its graph launches: one every ~0.75 ms per engine) on the CPUs of its GPU's NUMA
    node -- eight ranks x four engines launching from anywhere contend for the same cores and cross the socket link.
    Best effort: the PCI address of the device -> /sys/bus/pci/devices/<addr>/numa_node -> that node's cpulist; returns what it did."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return {"numa_node": node, "pinned": False, "why": "the platform reports no NUMA node for the device"}
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"numa_node": node, "pinned": False, "why": "none of the node's CPUs is in this process's affinity mask"}
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "pinned": True, "cpus": len(cpus), "pci": addr}
    except Exception as ex:   # (an unknown sysfs layout must not cost the run)
        return {"pinned": False, "why": f"{type(ex).__name__}: {ex}"}


def executed_flops_chain(A: int, e_a2p: float, e_m2p: float, layers: int) -> float:
    """FLOPs the factored kernel actually executes: per destination 26 D^2 + 2 D^2 (q~) + 2 D^2
    (to_v_r fold); per edge 2*(8*128) score + 2*(8*128) aggregate + 4 D."""
    per = lambda E: D * D * 30 * A + E * (4 * 8 * D + 4 * D)
    return layers * (per(e_a2p) + per(e_m2p))


def executed_mfma_flops_chain16(n_wg: int, tiles: float, layers: int) -> float:
    """FLOPs of the v_mfma_f32_16x16x32_f16 / 16x16x16_f16 instructions one k_chain16 policy launch issues (ps_chain16.h):
    per workgroup and layer the node GEMMs on 16-row tiles -- q|s|g 24 n-tiles x 4 k-blocks, q~ 48 x 1, to_v_r fold 8 x 3,
    gate 8 x 4, to_out 8 x 4, FFN up 32 x 4, FFN down 8 x 16, three split-fp16 products each = 1464 MFMAs of 16384 FLOP;
    per 16-edge tile 14 score MFMAs (16x16x32) + 12 aggregation MFMAs (16x16x16, 8192 FLOP).  `tiles` = sum over the
    layers' edge sets of ceil(deg / 16) per destination (one a2p + one m2p set per layer pair)."""
    node = n_wg * 2 * layers * 1464 * 16384.0
    edge = tiles * layers * (14 * 16384.0 + 12 * 8192.0)
    return node + edge


def cpu_baseline(spec, w, scene, reps: int = 3):
    """The oracle (CPU restatement, oracle/prosim_oracle.py) timed on this box's host cores."""
    from oracle import prosim_oracle as orc
    phys = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or phys
    except Exception:
        pass
    # torch's CPU kernels on these small graphs stop scaling (and then regress) well before all cores:
    # sweep a few thread counts once and time the best one -- the strongest CPU number we can produce
    best, cores = None, 1
    with torch.no_grad():
        for nt in sorted({c for c in (8, 16, 32, phys) if c <= phys}):
            torch.set_num_threads(nt)
            orc.rollout(w, spec, scene)  # warm-up at this thread count
            t0 = time.perf_counter()
            orc.rollout(w, spec, scene)
            dt_ = time.perf_counter() - t0
            if best is None or dt_ < best:
                best, cores = dt_, nt
        torch.set_num_threads(cores)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.rollout(w, spec, scene)
            ts.append(time.perf_counter() - t0)
    A = int(scene["prompt_mask"].sum())
    med = float(np.median(ts))
    # the reference itself cannot travel to this box; its time on the BUILD container's cores, measured there beside the port
    # (tools/time_reference_cpu.py), rides along for context
    ref_ctx = None
    rp = os.path.join(ROOT, "profiles", "r04_reference_cpu_time.json")
    if os.path.exists(rp):
        with open(rp) as f:
            rj = json.load(f)
        ref_ctx = {"source": "profiles/r04_reference_cpu_time.json (tools/time_reference_cpu.py, build container, NOT this box)", "threads": rj["threads"],
                   "reference_plus_standins_s_per_rollout": rj["reference_plus_standins"]["best"],
                   "reference_plus_standins_agent_steps_per_s": rj["reference_plus_standins"]["agent_steps_per_s"],
                   "oracle_port_s_per_rollout_same_cores": rj["oracle_port"]["best"]}
    return dict(value=A * spec.max_steps / med, unit="agent-steps/s", cores=cores, kind="port", reference_in_build_container=ref_ctx,
                sample=f"{reps} full rollouts of ONE scene of the batch (CPU throughput does not depend on the batch; median {med:.3f} s each), torch {torch.__version__} fp32, "
                       f"{cores} intra-op threads (best of a sweep over 8/16/32/all {phys} physical cores)")


def run_empty_rank(args, spec, n_scenes, world, backend, dev_index, multi):
    """A rank that owns no scene (more ranks than scenes): it still takes part in every collective of the job -- the gather's
    set-up, the two metric computations, the barriers and the max-over-ranks / agent-count reductions -- with zero rows."""
    from prosim_amd.distributed import SceneMetricGather
    assert multi, "a rank without scenes exists only in a multi-rank job"
    N = synth.BASELINE_CONFIGS[args.config]["n_agents"]
    dev = "cuda" if backend == "nccl" else "cpu"
    gather = SceneMetricGather([], n_scenes, N, 10, dev)
    empty = torch.zeros(0, N, 10, device=dev)
    gather(empty)                      # (warm-up computation)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gather(empty)                      # the timed region's metric computation
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    agents = torch.tensor([0], device=dev, dtype=torch.float64)
    dist.all_reduce(agents)
    # the per-rank diagnostic of the ranks that own scenes (main: `per_rank`): the same collectives, nothing to roll out
    dist.barrier()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gather(empty)
    torch.cuda.synchronize()
    mine = torch.tensor([0.0, 1e3 * (time.perf_counter() - t2)], device=dev, dtype=torch.float64)
    dist.all_gather([torch.zeros_like(mine) for _ in range(world)], mine)
    gather(empty)                      # the final metric computation after the event-timed rollouts
    dist.barrier()
    dist.destroy_process_group()


def trace(msg: str):
    """PS_BENCH_TRACE=1: progress marks on stderr (which section of rank 0's extra measurements a fault belongs to)."""
    if os.environ.get("PS_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps; the pipeline of rollouts in flight fills and drains once inside the timed region (~10 ms), so short runs "
                         "under-report the steady state: 5 steps 15.0 M, 10: 18.7 M, 20: 21.1 M, 100: 21.2 M, 200: 21.3 M agent-steps/s")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scenes-per-gpu", type=int, default=8)
    ap.add_argument("--total-scenes", type=int, default=0,
                    help="job size in scenes when it is NOT scenes-per-gpu x ranks (uneven shards, ranks without a scene: the sharding is i %% world); "
                         "0 = scenes-per-gpu x ranks")
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config index used as the per-scene workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-rows", type=int, default=-1, choices=[-1, 0, 1, 2, 4, 8, 9, 10, 11, 12, 13, 14, 15, 16],
                    help="rows per workgroup of the fused attention launches (ps_set_chain_rows); -1: 16 when rollouts are pipelined, else 0")
    ap.add_argument("--inflight", type=int, default=16,
                    help="rollouts in flight per GPU: consecutive steps alternate between this many engines (own buffers and "
                         "stream each) that hold the same resident batch, so step k+1 starts while step k drains.  Multiples of 4 "
                         "(4 policy launches of 64 workgroups fill the 256 CUs); the line reports 4 and 8 beside the default")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the rollout path has no CPU fallback")
    # test hook: PS_BENCH_BACKEND=gloo PS_BENCH_SAME_DEVICE=1 runs an N-rank job on ONE GPU (both ranks on
    # cuda:0, metric gather through gloo on CPU copies) to exercise the N > 1 code path on a 1-GPU box
    backend = os.environ.get("PS_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("PS_BENCH_SAME_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    numa = pin_to_gpu_numa_node(dev_index) if (world > 1 and "LOCAL_RANK" in os.environ and not os.environ.get("PS_BENCH_NO_PIN")) else None
    # test hook: PS_BENCH_FORCE_DIST=1 takes the N > 1 code path (process group, RCCL gather at metric-compute time) with ONE
    # rank, so the collective path can be exercised with the real nccl backend on a 1-GPU box
    multi = world > 1 or bool(os.environ.get("PS_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if multi:
        dist.barrier()
    from prosim_amd.engine import Engine
    from prosim_amd.distributed import shard_scenes, reduce_pair_metrics, rows_to_slots

    spec = DEMO_SPEC
    w = weights.init_weights(spec, 0)
    n_scenes = args.total_scenes if args.total_scenes > 0 else args.scenes_per_gpu * world
    my_scenes = shard_scenes(n_scenes, rank, world)
    S = len(my_scenes)            # this rank's scenes (== --scenes-per-gpu unless --total-scenes makes the shards uneven)
    if S == 0:
        return run_empty_rank(args, spec, n_scenes, world, backend, dev_index, multi)
    # scene i of the job is seed i of the generator; this rank's batch = its shard, in shard order
    parts = [synth.baseline_scene(spec, args.config, seed=i, batch=1) for i in my_scenes]
    scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
                 {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]})
             for k in parts[0]}
    # Steps are independent rollouts of the resident batch, so consecutive steps are PIPELINED: `inflight` engines (each
    # with its own device buffers and non-blocking stream, all holding the same batch) take the steps in turn; the
    # tail of one rollout (the latency-bound last replans of its slowest workgroups) overlaps the head of the next.
    # Every step is still a complete rollout of the whole batch inside the timed region.
    n_fl = max(1, args.inflight)
    engines = [Engine(spec, w, device=dev_index) for _ in range(n_fl)]
    chain_rows = args.chain_rows if args.chain_rows >= 0 else (16 if n_fl > 1 else 0)
    for e_ in engines:
        e_.set_chain_rows(chain_rows)
        e_.set_scene(scene)
    eng = engines[0]
    A = eng.num_agents
    N = scene["prompt_mask"].shape[1]
    from prosim_amd.distributed import SceneMetricGather
    # Metrics follow the reference's torchmetrics flow (metrics/motion_pred.py:111-199 under Lightning): every step UPDATES
    # the metric state on the device (ps_pair_metric: per-agent sums and counts, enqueued behind the rollout on the
    # engine's stream), and the cross-rank exchange happens ONCE, when the metric is computed -- one RCCL all-gather of
    # the per-scene rows at the end of the timed region, inside it.  It is enqueued on the stream of the engine that ran
    # the last step: a cross-stream event wait behind a graph launch costs ~1 ms of pipeline on ROCm 7.2
    # (tools/gpu_gather_cost.py: 4.11 -> 5.1 ms per step with one such wait per step, 4.11 with none).
    NM = 10                                              # floats per agent row of ps_pair_metric
    metric_bufs = [torch.zeros(A, NM, device="cuda") for _ in range(n_fl)]
    eng_streams = [torch.cuda.ExternalStream(e_.stream_handle, device=torch.device("cuda", dev_index)) for e_ in engines]
    gather = SceneMetricGather(my_scenes, n_scenes, N, NM, "cuda" if backend == "nccl" else "cpu") if multi else None
    state = {"k": 0, "last": None}
    # seeded synthetic ground truth of the metric (there is no log behind synthetic scenes): per (replan, agent row) local
    # targets with gaps, scene i of the job = generator seed i, resident on the device like the scenes themselves
    R_, S_ = spec.n_replans, spec.target_steps
    slots_np = eng.row_slots
    tg_parts = [synth.make_pair_metric_inputs(i, B=1, N=N, R=R_, K=1, S=S_) for i in my_scenes]
    tgt_all = np.concatenate([p["tgt"] for p in tg_parts]).transpose(1, 0, 2, 3, 4).reshape(R_, S * N, S_, 5)
    msk_all = np.concatenate([p["mask"] for p in tg_parts]).transpose(1, 0, 2).reshape(R_, S * N)
    t_tgt = torch.from_numpy(np.ascontiguousarray(tgt_all[:, slots_np])).cuda()
    t_msk = torch.from_numpy(np.ascontiguousarray(msk_all[:, slots_np].astype(np.uint8))).cuda()
    slots = torch.from_numpy(slots_np).cuda()
    torch.cuda.synchronize()   # (these tensors were filled on torch's default stream and are read on the engines' non-blocking streams: order them once)

    def step():
        i = state["k"] % state.get("n_fl", n_fl)
        state["k"] += 1
        engines[i].rollout()
        engines[i].pair_metric(metric_bufs[i].data_ptr(), t_tgt.data_ptr(), t_msk.data_ptr())
        state["last"] = (i, None)

    def compute_metrics():
        """PairMotionPred.compute(): the per-scene rows of the last update, from every rank, in scene order."""
        i, _ = state["last"]
        if not multi:
            with torch.cuda.stream(eng_streams[i]):
                lg = rows_to_slots(metric_bufs[i], slots, S, N)
        elif backend == "nccl":
            with torch.cuda.stream(eng_streams[i]):
                lg = gather(rows_to_slots(metric_bufs[i], slots, S, N))
        else:   # CPU test hook: the copy to the host is the wait
            engines[i].sync()
            lg = gather(rows_to_slots(metric_bufs[i], slots, S, N).cpu())
        state["last"] = (i, lg)

    # set-up, not steps: every engine captures its rollout graph once (the first rollout of an engine records ~330 launches into a
    # hipGraph; with fewer warm-up steps than engines that capture would otherwise land inside the timed region)
    for e_ in engines:
        e_.rollout()
    for e_ in engines:
        e_.sync()
    for _ in range(args.warmup):
        step()
    if state["last"] is None:   # --warmup 0
        step()
    compute_metrics()   # (warm: the first collective builds RCCL's communicator)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    compute_metrics()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        agents = torch.tensor([A], device=red_dev, dtype=torch.float64)
        dist.all_reduce(agents)
        total_agents = int(agents.item())
    else:
        total_agents = A
    ms_per_step = 1e3 * dt / args.steps
    graph_nodes = engines[0].graph_nodes   # (read while engine 0 still holds the graph it replayed in the timed region)
    per_rank = None
    if multi:
        # Outside the timed region, for the reader of an N > 1 line: every rank's OWN step time with no collective in the loop
        # (host launch contention between the ranks of a node shows up here) and what one metric all-gather costs it (the
        # collective itself; the ranks enter it together).  A shortfall from N x the 1-GPU value is one or the other.
        for e_ in engines:
            e_.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        for e_ in engines:
            e_.sync()
        local_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        compute_metrics()
        engines[state["last"][0]].sync()
        torch.cuda.synchronize()
        mine = torch.tensor([local_ms, 1e3 * (time.perf_counter() - t2)], device=red_dev, dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = {"host_affinity_rank0": numa,
                    "local_ms_per_step_no_collective": [float(v[0]) for v in allv], "metric_all_gather_ms": [float(v[1]) for v in allv],
                    "note": "measured after the timed region; a rank without a scene reports 0 for its step time"}
    # launch durations of the dominant kernel while the pipeline is full: the same loop again for 2 rounds of the engines with
    # an event pair around every policy launch (such rollouts are launched eagerly -- events do not survive graph replay
    # on ROCm 7.2 -- at ~1 ms of host time each against a 6 ms step), read after the last one
    trace("timed region done; event-timed policy launches")
    # ... at FOUR rollouts in flight: four 64-workgroup launches have the 256 CUs to themselves.  Deeper than that (the timed region runs 16) a
    # launch's workgroups queue for CUs behind other engines' launches and the event pair -- like rocprofv3's kernel duration -- times the queue
    # (ev_full_ms below: reported, not used for the roofline).
    n_ev = min(n_fl, 4)
    for e_ in engines:
        e_.sync()
    state["n_fl"] = n_ev
    for e_ in engines[:n_ev]:
        e_.enable_policy_events(True)
    for _ in range(2 * n_ev):
        step()
    ev_ms = np.concatenate([e_.policy_event_times() for e_ in engines[:n_ev]])
    state.pop("n_fl", None)
    for e_ in engines:
        e_.sync()
        e_.enable_policy_events(True)
    for _ in range(2 * n_fl):
        step()
    ev_full_ms = np.concatenate([e_.policy_event_times() for e_ in engines])
    for e_ in engines:
        e_.enable_policy_events(False)
    value = total_agents * spec.max_steps / (dt / args.steps)
    for e_ in engines:
        e_.sync()
    # the same loop with fewer rollouts in flight (this rank's own rate, no collective): what the depth of the pipeline is worth
    by_inflight = {str(n_fl): A * spec.max_steps / (dt / args.steps) if not multi else None}
    for n_sub in (4, 8):
        if n_sub < n_fl:
            state["n_fl"] = n_sub
            for _ in range(2 * n_sub):
                step()
            for e_ in engines:
                e_.sync()
            t_sub = time.perf_counter()
            for _ in range(max(32, args.steps // 2)):
                step()
            for e_ in engines:
                e_.sync()
            by_inflight[str(n_sub)] = A * spec.max_steps * max(32, args.steps // 2) / (time.perf_counter() - t_sub)
    state.pop("n_fl", None)
    free_b, total_b = torch.cuda.mem_get_info()
    hbm_in_use_gb = (total_b - free_b) / 1e9   # (the n_fl engines of the timed region with their captured graphs, torch's pools, the metric state)
    step()
    compute_metrics()
    for e_ in engines:
        e_.sync()
    torch.cuda.synchronize()
    li, lg = state["last"]
    # one PairMotionPred update per rank (its scenes are its batch, rollout/callbacks.py:76), MeanMetric over the updates
    metrics = reduce_pair_metrics(lg, batches=[shard_scenes(n_scenes, r_, world) for r_ in range(world)])
    metrics["scenes"] = int(n_scenes)
    for e_ in engines[1:]:   # the per-stage / per-kernel timings below run on one engine, alone on the GPU
        e_.close()

    if rank == 0:
        # dominant kernel: the fused policy attention chain (one launch per replan), timed with HIP
        # events on the engine's own stream
        trace("rank-0 extras: policy kernel alone")
        ms_chain = eng.time_policy_kernel(3)
        ec = eng.get("edge_counts")
        ms_roll, stages = eng.time_rollout(1, 5)
        # one rollout alone on the GPU in the engine's latency mode (2 rows per workgroup, two workgroups per CU)
        trace("latency mode")
        eng.set_chain_rows(0)
        ms_roll_lat, stages_lat = eng.time_rollout(1, 5)
        ms_chain_lat = eng.time_policy_kernel(3)
        eng.set_chain_rows(chain_rows)
        # single-scene latency of the same workload (S = 1), same engine, same run, latency mode
        eng.set_chain_rows(0)
        trace("single scene")
        eng.set_scene(parts[0])
        eng.rollout(); eng.sync()
        graph_nodes_single = eng.graph_nodes
        ms_single, stages1 = eng.time_rollout(2, 10)
        ms_chain1 = eng.time_policy_kernel(2)
        A1 = eng.num_agents
        # ... and the throughput of that single-scene workload (BASELINE configs[2]) when its rollouts are pipelined:
        # 128 agents are 128 workgroups on 256 CUs (two of them fit a CU), so rollouts in flight fill the rest of the chip;
        # beyond two in flight the host's graph launches (one per ~2 ms rollout) become the limit
        pipe1 = {}
        trace("single scene pipelined")
        for nfl1 in (2, 6):
            es1 = [Engine(spec, w, device=dev_index) for _ in range(nfl1)]
            for e_ in es1:
                e_.set_scene(parts[0])
                e_.rollout()
            for e_ in es1:
                e_.sync()
            t1 = time.perf_counter()
            for k_ in range(10 * nfl1):
                es1[k_ % nfl1].rollout()
            for e_ in es1:
                e_.sync()
            pipe1[str(nfl1)] = A1 * spec.max_steps * 10 * nfl1 / (time.perf_counter() - t1)
            for e_ in es1:
                e_.close()
        # the Sim-Agents shape: ONE scene x 32 replicas, parallel_rollout_batch (rollout/gpu_utils.py:179-228).
        # ps_set_replicas computes the scene encoder and the generator once and keeps the map once; the reference's way
        # (the 32-fold replicated batch through the whole path) runs on the same engine beside it.
        trace("replicas")
        MREP = 32
        eng.set_chain_rows(0)
        eng.set_replicas(MREP)
        eng.set_scene(parts[0])
        ms_rep, st_rep = eng.time_rollout(1, 3)
        A_rep = eng.num_agents
        # its policy launch has 4096 rows = 256 workgroups of 16: ONE launch fills the chip, so here the per-launch rate is
        # the chip's rate (the 8-scene launch of the headline has 64 workgroups and shares the chip with three others)
        ms_chain_rep = eng.time_policy_kernel(3)
        ec_rep = eng.get("edge_counts")
        fl_rep = algorithmic_flops_chain(A_rep, float(ec_rep[4]), float(ec_rep[5]), spec.pol_layers)
        world_buf = torch.zeros(A_rep, spec.max_steps, 3, device="cuda")
        tf = np.array([[0.6, -0.8, 3418.7], [0.8, 0.6, -1650.2], [0, 0, 1]], np.float32)
        eng.world_trajs(tf, world_buf.data_ptr())
        eng.sync()
        t_w = time.perf_counter()
        for _ in range(20):
            eng.world_trajs(tf, world_buf.data_ptr())
        eng.sync()
        ms_world = 1e3 * (time.perf_counter() - t_w) / 20
        trace("replicated batch")
        eng.set_replicas(1)
        eng.set_scene(replicate_scene(parts[0], MREP))
        ms_tiled, st_tiled = eng.time_rollout(1, 3)
        replica = {"workload": f"1 x BASELINE configs[{args.config}] scene x {MREP} replicas (ps_set_replicas: shared encoder + generator, one copy of the map)",
                   "agent_rows": A_rep, "ms_per_rollout": ms_rep, "agent_steps_per_s": A_rep * spec.max_steps / (ms_rep * 1e-3),
                   "stage_ms": {"encode_scene": st_rep[0], "generate_policy": st_rep[1], "replan_loop": st_rep[2]},
                   "replicated_batch": {"note": "the reference's layout: every tensor .repeat(32) on the batch dim, whole path per replica",
                                        "ms_per_rollout": ms_tiled,
                                        "stage_ms": {"encode_scene": st_tiled[0], "generate_policy": st_tiled[1], "replan_loop": st_tiled[2]}},
                   "policy_chain_launch": {"rows": A_rep, "workgroups": (A_rep + 15) // 16, "ms": ms_chain_rep,
                                           "algorithmic_flops": fl_rep, "algorithmic_tflops": fl_rep / (ms_chain_rep * 1e-3) / 1e12,
                                           "frac_of_f16_mfma_peak": fl_rep / (ms_chain_rep * 1e-3) / 1e12 / 2500.0},
                   "speedup_vs_replicated_batch": ms_tiled / ms_rep,
                   "world_frame_kernel_ms": ms_world}
        # the headline's honest sibling: a stream of NEW batches (RolloutPipeline, throughput mode): every step uploads a fresh
        # 8-scene batch from host arrays (ps_set_scene), captures and runs its rollout and reads traj / vel back
        eng.set_replicas(1)
        from prosim_amd.stream import RolloutPipeline
        new_batches = [synth.baseline_scene(spec, args.config, seed=1000 + i, batch=S) for i in range(6)]
        streaming = {"workload": f"stream of NEW {S}-scene batches (host arrays -> ps_set_scene -> rollout -> traj / vel read back), RolloutPipeline, "
                                 "16 rows per workgroup from depth 2", "agent_steps_per_s_by_depth": {}, "ms_per_batch_by_depth": {}}
        for depth in (1, 2, 4, 8, 16):
            trace(f"streaming depth {depth}")
            with RolloutPipeline(spec, w, device=dev_index, depth=depth) as pipe:
                for _ in pipe.run((new_batches * ((depth + 5) // 6))[:depth]):   # (every engine once: its rollout graph is captured outside the timed region)
                    pass
                t_s = time.perf_counter()
                # (the pipeline's fill and drain -- 2 x depth batches deep, one rollout latency each way -- stay a small part of the timed region)
                n_b = sum(1 for _ in pipe.run(new_batches * (8 if depth < 3 else 6 * depth)))
                dt_s = time.perf_counter() - t_s
            streaming["agent_steps_per_s_by_depth"][str(depth)] = n_b * A * spec.max_steps / dt_s
            streaming["ms_per_batch_by_depth"][str(depth)] = 1e3 * dt_s / n_b
        streaming["agent_steps_per_s"] = max(streaming["agent_steps_per_s_by_depth"].values())
        trace("tiles of the last replan")
        fl_alg = algorithmic_flops_chain(A, float(ec[4]), float(ec[5]), spec.pol_layers)
        # per-destination degrees of the last replan's edge sets -> 16-edge tiles the edge phase walked
        eng.set_chain_rows(chain_rows)
        eng.set_scene(scene)
        eng.rollout()
        tiles = 0.0
        for which in (4, 5):
            _, edst, _ = eng.get_edges(which, cap=1 << 22)
            tiles += float(np.ceil(np.bincount(edst, minlength=A) / 16.0).sum())
        c16 = chain_rows >= 8
        n_wg = (A + chain_rows - 1) // chain_rows if c16 else 0
        fl_mfma = executed_mfma_flops_chain16(n_wg, tiles, spec.pol_layers) if c16 else None
        ms_launch = float(ev_ms.mean())
        peak = 2500.0   # TFLOP/s: dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md) -- the instruction class the kernel issues
        # HBM-side traffic of the launch comes from separate rocprofv3 --pmc passes (tools/gpu_round_profile.sh; counters
        # cannot be read from inside this process): offline, valid for the default workload only, stamped with its source
        traffic, traffic_src = None, None
        pmc = newest_pmc_json()
        if pmc and S == 8 and args.config == 2:
            with open(pmc) as f:
                pj = json.load(f)
            if pj.get("chain_rows") == chain_rows:
                traffic = pj["hbm_bytes_per_launch"]
                traffic_src = {"file": os.path.relpath(pmc, ROOT), "git": pj.get("git"), "measured": "offline rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), not in this run"}
        Mv_b = int(np.asarray(scene["map_mask"]).any(-1).sum())   # map tokens with a valid point: the m2p sources
        alg_bytes = algorithmic_bytes_chain(A, A, Mv_b, float(ec[4]), float(ec[5]), spec.pol_layers)
        achieved = fl_alg / (ms_launch * 1e-3) / 1e12
        kern = (f"k_chain16<8, policy> (12 fused attention layers per launch, {chain_rows} rows per 8-wave workgroup, {n_wg} workgroups)" if c16
                else f"k_attn_chain (policy: 12 fused attention layers per launch; {chain_rows or 2} rows per workgroup)")
        out = {
            "metric": "agent-steps/sec closed-loop rollout", "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # fp32 throughout; the matrix-core GEMMs take split-fp16 operands with fp32 accumulation (fp32-class)
            "dtype": "f32 (split-f16x3 MFMA: fp16 hi + lo operands, three products per multiply, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{S} x BASELINE configs[{args.config}] scenes per GPU ({synth.BASELINE_CONFIGS[args.config]['name']}; "
                                   f"S=8 is configs[3]'s per-GPU share), 80-step closed-loop rollout (8 replans), seeded random-init weights",
                       "agents_per_scene": int(scene['prompt_mask'][0].sum()), "polylines_per_scene": int(scene['map_mask'].shape[1]),
                       "scenes_per_gpu": S, "scenes_total": n_scenes, "rollouts_in_flight": n_fl, "hbm_in_use_gb_rank0": hbm_in_use_gb, "agent_steps_per_s_by_rollouts_in_flight_rank0": by_inflight,
                       "chain_rows_per_workgroup": chain_rows,
                       "parallelism": f"scene-sharded x{world}, RCCL all-gather of the per-agent PairMotionPred sums; consecutive steps pipelined over "
                                      f"{n_fl} engine(s) per GPU"},
            # bound: what the counters of the launch say (profiles/r03_*_pmc_chain16.txt: the VALU busy more than half of the SIMD
            # cycles, the waves parked a third of theirs, MFMA 13 %, HBM < 5 %); peak / frac stay on the dense f16 MFMA yardstick
            # that BASELINE.json's north_star names
            "roofline": {"bound": "valu-issue/latency", "yardstick": "mfma", "kernel": kern,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "note": "achieved = ALGORITHMIC FLOPs of the reference formulation per launch (SURVEY.md section 8(d): per-edge to_k_r / "
                                 "to_v_r GEMVs counted) / avg_launch_ms; peak = dense f16 MFMA, the instruction class the kernel issues "
                                 "(v_mfma_f32_16x16x32_f16 on split-fp16 operands).  The kernel factors the per-edge GEMVs out and executes "
                                 "fewer FLOPs: executed_mfma_* is the hardware-side figure (instruction counts of ps_chain16.h x FLOPs per "
                                 "instruction; the PMC pass SQ_INSTS_VALU_MFMA_MOPS_F16 under profiles/ counts the same instructions).  "
                                 "A launch occupies `workgroups` of the 256 CUs and `rollouts_in_flight` launches overlap, so the per-launch "
                                 "rate understates the chip: chip_algorithmic_tflops = all policy launches of the timed region / its wall time.  "
                                 "What bounds the launch: DESIGN.md section 0a / 4 (edge phase: the dependent LDS / MFMA / cross-lane chain of a 16-edge tile at two waves per SIMD "
                                 "over a VALU floor that round 5 cut from 448 to 348 static instructions per tile -- 51.6 M -> 41.1 M VALU instructions per launch for 5 % of its time, "
                                 "so issue count is no longer the first limiter; node phase: the dependent stage chain of a 16-row layer -- GEMM, LDS, barrier, epilogue; 9 barriers per layer since round 6 (15 before), the elementwise steps on the GEMMs' accumulators, a layer's 960 KB of weight fragments as one compile-time stream through a ring of three register slots at 64 B/clk per CU).",
                         "algorithmic_bytes": alg_bytes["total"], "algorithmic_bytes_parts": alg_bytes,
                         "traffic_over_algorithmic_bytes": (traffic / alg_bytes["total"]) if traffic else None,
                         "algorithmic_flops_per_launch": fl_alg,
                         "executed_mfma_flops_per_launch": fl_mfma,
                         "executed_mfma_tflops": (fl_mfma / (ms_launch * 1e-3) / 1e12) if fl_mfma else None,
                         "executed_mfma_frac": (fl_mfma / (ms_launch * 1e-3) / 1e12 / peak) if fl_mfma else None,
                         "chip_algorithmic_tflops": fl_alg * spec.n_replans / (ms_per_step * 1e-3) / 1e12,
                         "hbm_gbps": (traffic / (ms_launch * 1e-3) / 1e9) if traffic else None, "hbm_peak_gbps": 8000.0,
                         "avg_launch_ms": ms_launch,
                         "launch_timing": f"HIP event pairs around the policy launches on each engine's stream, in the same pipelined loop as the timed "
                                          f"region, run right after it with {min(n_fl, 4)} rollouts in flight (four 64-workgroup launches share the 256 CUs without queueing; launched "
                                          f"eagerly: events do not survive graph replay): the {len(ev_ms)} launches of each engine's last rollout (min "
                                          f"{float(ev_ms.min()):.3f} / max {float(ev_ms.max()):.3f} ms); alone on the GPU the same launch takes launch_alone_ms; with all "
                                          f"{n_fl} rollouts of the timed region in flight a launch's workgroups wait for CUs and the same event pairs read "
                                          f"avg_launch_ms_with_cu_queueing",
                         "avg_launch_ms_with_cu_queueing": float(ev_full_ms.mean()),
                         "launch_alone_ms": ms_chain,
                         "workgroups": n_wg if c16 else None,
                         "occupied_cus": min(n_wg, 256) if c16 else None,
                         "chip_frac": fl_alg * spec.n_replans / (ms_per_step * 1e-3) / 1e12 / peak,
                         "full_chip_launch_frac": fl_rep / (ms_chain_rep * 1e-3) / 1e12 / peak,
                         "edges_per_launch": {"a2p": float(ec[4]), "m2p": float(ec[5])}, "tiles16_per_layer_pair": tiles},
            "graph_nodes_per_rollout": graph_nodes,
            "stage_ms": {"rollout_events": ms_roll, "encode_scene": stages[0], "generate_policy": stages[1], "replan_loop": stages[2]},
            "latency_mode": {"note": "ps_set_chain_rows(0): one rollout alone on the GPU", "ms_per_rollout": ms_roll_lat,
                             "encode_scene": stages_lat[0], "generate_policy": stages_lat[1], "replan_loop": stages_lat[2],
                             "policy_chain_launch_ms": ms_chain_lat},
            "single_scene": {"ms_per_rollout": ms_single, "agent_steps_per_s": A1 * spec.max_steps / (ms_single * 1e-3),
                             "policy_chain_launch_ms": ms_chain1, "graph_nodes_per_rollout": graph_nodes_single,
                             "agent_steps_per_s_pipelined": pipe1,   # key = rollouts in flight
                             "stage_ms": {"encode_scene": stages1[0], "generate_policy": stages1[1], "replan_loop": stages1[2]}},
            "replica_fanout": replica,
            "streaming": streaming,
            "rollout_metrics": metrics,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, w, parts[0])
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        trace("line")
        print(json.dumps(out), flush=True)
    eng.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
