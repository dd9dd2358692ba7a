"""Round 5: serving a stream of NEW batches -- the captured hipGraph is kept across scenes of one shape (ps_rollout compares the
signature of its launch sequence instead of re-capturing after every setter), uploads go through pinned staging without a stream
synchronisation, results are copied back behind the rollout (ps_prefetch).  Every shortcut must return the bits of a fresh engine."""
import numpy as np
import pytest

from prosim_amd import synth, weights
from prosim_amd.spec import SMALL_SPEC

pytestmark = pytest.mark.gpu


def _fresh(spec, w, scene, rows):
    from prosim_amd.engine import Engine
    eng = Engine(spec, w)
    eng.set_chain_rows(rows)
    eng.set_scene(scene)
    eng.rollout()
    out = (eng.padded("traj"), eng.get("motion_pred"), eng.padded("vel"))
    eng.close()
    return out


@pytest.mark.parametrize("rows", [0, 16])
def test_graph_is_kept_across_scenes_of_one_shape_and_recaptured_when_the_shape_moves(rows):
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    same = [synth.make_scene(spec, 12, 40, batch=2, seed=70 + i, goal=True) for i in range(3)]       # one shape, three contents
    other = synth.make_scene(spec, 9, 33, batch=3, seed=80, goal=True, ragged=True)                   # another shape
    nogoal = synth.make_scene(spec, 12, 40, batch=2, seed=81, goal=False)                             # the shape of `same` without conditions
    eng = Engine(spec, w)
    eng.set_chain_rows(rows)
    seq = [same[0], same[1], same[2], other, same[0], nogoal, same[1]]
    for k, sc in enumerate(seq):
        eng.set_scene(sc)
        eng.rollout()
        got = (eng.padded("traj"), eng.get("motion_pred"), eng.padded("vel"))
        want = _fresh(spec, w, sc, rows)
        for g, x in zip(got, want):
            assert np.array_equal(g, x), (rows, k)
    cap, reuse = eng.graph_stats()
    # captures: same[0], other, same[0] again (buffers may have grown for `other`: then its pointers moved), nogoal (no condition
    # layers), same[1] (conditions back); the two batches right behind same[0] replay its graph
    assert reuse >= 2, (cap, reuse)
    assert cap + reuse == len(seq), (cap, reuse)
    assert cap >= 3, (cap, reuse)
    # a second rollout of the resident scene neither captures nor counts as a re-check
    eng.rollout()
    assert eng.graph_stats() == (cap, reuse)
    eng.close()


def test_get_async_delivers_the_bits_of_get_and_a_batch_can_queue_behind_a_rollout_in_flight():
    import torch
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    a = synth.make_scene(spec, 10, 30, batch=2, seed=90, goal=True, tags=True)
    b = synth.make_scene(spec, 10, 30, batch=2, seed=91, goal=True, tags=True)
    names = ("traj", "vel", "motion_pred", "policy_emd", "fused")
    want = {}
    for key, sc in (("a", a), ("b", b)):
        eng = Engine(spec, w)
        eng.set_scene(sc)
        eng.rollout()
        want[key] = {n: eng.get(n) for n in names}
        eng.close()
    eng = Engine(spec, w)
    bufs = {}
    for key, sc in (("a", a), ("b", b)):   # b is uploaded and launched while a's rollout may still be running: no sync in between
        eng.set_scene(sc)
        eng.rollout()
        for n in names:
            shp = eng.result_shape(n)
            t = torch.empty(int(np.prod(shp)), dtype=torch.float32, pin_memory=True)
            assert eng.get_async(n, t.data_ptr(), t.numel()) == t.numel()
            bufs[(key, n)] = (t, shp)
    eng.sync()
    for (key, n), (t, shp) in bufs.items():
        assert np.array_equal(t.numpy().reshape(shp), want[key][n]), (key, n)
    assert eng.graph_stats() == (1, 1)
    with pytest.raises(RuntimeError, match="not a per-agent result"):
        eng.get_async("scene_tokens", bufs[("a", "traj")][0].data_ptr(), 16)
    with pytest.raises(RuntimeError, match="too small"):
        eng.get_async("traj", bufs[("a", "traj")][0].data_ptr(), 16)
    eng.close()


def test_a_batch_type_without_conditions_right_after_one_with_them_runs_without_the_condition_layers():
    """Engine.set_scene skips the condition setters for types the batch does not carry (ps_set_scene cleared them): the result must
    be that of an engine that never saw a condition."""
    from prosim_amd.engine import Engine
    spec = SMALL_SPEC
    w = weights.init_weights(spec, 0)
    withc = synth.make_scene(spec, 8, 24, batch=2, seed=95, goal=True, tags=True)
    without = synth.make_scene(spec, 8, 24, batch=2, seed=96, goal=False)
    eng = Engine(spec, w)
    eng.set_scene(withc)
    eng.rollout()
    eng.set_scene(without)
    eng.rollout()
    got = eng.padded("traj")
    eng.close()
    assert np.array_equal(got, _fresh(spec, w, without, 0)[0])


# ------------------------------------------------------------------ the REFERENCE itself at the other BASELINE sizes (missing 2)
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_REF_FULL = {   # fixture -> (spec overrides, baseline config, batch, the workload whose cut-agent list applies)
    "demo_cfg1_seed0": ({}, 1, None, "baseline_configs/cfg1_seed0"),
    "demo_cfg3_seed0_b2": ({}, 3, 2, "baseline_configs/cfg3_seed0"),
    "demo_cfg4_seed0": ({}, 4, None, "baseline_configs/cfg4_seed0"),
    "demo_cfg4_notrunc": (dict(dec_max_neigh=1280, pol_max_neigh=1280), 4, None, "no_truncation/cfg4"),
}


@pytest.mark.parametrize("name", sorted(_REF_FULL))
def test_engine_against_the_reference_made_fixture_at_full_size(name):
    """configs[1], configs[3] seed 0's two scenes, configs[4] and its no-truncation variant against the REFERENCE's own fp32 forward
    (tests/gen_golden.py FULL_CASES, round 5) -- until now these sizes were engine-vs-oracle only.  Replan 0 within 1e-4 of the
    reference; closed loop: an agent further than 1e-4 from the reference must be one where an fp32 run is known to toss a coin --
    on the workload's committed cut list, or a row where the REFERENCE's own run left the fp64 oracle's trajectory.  The table
    records, per fixture, which side the reference landed on at the listed rows."""
    from prosim_amd.engine import Engine
    from prosim_amd.spec import DEMO_SPEC
    from parity_table import record, per_agent, known_cut_agents
    over, cfg, batch, workload = _REF_FULL[name]
    spec = DEMO_SPEC.replace(**over) if over else DEMO_SPEC
    w = weights.init_weights(spec, 0)
    scene = synth.baseline_scene(spec, cfg, seed=0, batch=batch)
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    pm = scene["prompt_mask"].astype(bool)
    A = int(pm.sum())
    ref_err = g["ref_err_per_agent"]                       # the reference's fp32 run against the fp64 oracle, per policy agent
    assert ref_err.shape == (A,)
    eng = Engine(spec, w)
    try:
        eng.set_scene(scene)
        eng.rollout()
        mp = eng.get("motion_pred")
        e0 = float(np.abs(mp[0] - g["motion_pred"][:A]).max())
        d = np.abs(eng.padded("traj") - g["traj"])[pm].reshape(A, -1).max(1)
    finally:
        eng.close()
    listed = known_cut_agents(workload)
    ref_out = set(int(i) for i in np.nonzero(ref_err >= 1e-4)[0])
    eng_out = set(int(i) for i in np.nonzero(d >= 1e-4)[0])
    record(f"vs_reference_full_size/{name}", replan0_max=e0, reference_outside_1e4_of_fp64=sorted(ref_out),
           reference_err_at_listed_rows={str(a): float(ref_err[a]) for a in sorted(listed)},
           engine_vs_reference_at_listed_rows={str(a): float(d[a]) for a in sorted(listed)}, **per_agent(d))
    print(f"{name}: replan-0 vs reference {e0:.2e} | per agent vs reference median {np.median(d):.2e} max {d.max():.2e} | outside 1e-4: "
          f"engine-vs-reference {sorted(eng_out)}, reference-vs-fp64 {sorted(ref_out)}, listed {sorted(listed)}")
    assert e0 < 1e-4
    assert eng_out <= (listed | ref_out), (name, sorted(eng_out - listed - ref_out))
    # (rows where the REFERENCE's own fp32 run left the fp64 trajectory -- eight of configs[3] seed 0's 256 agents, up to 3.4e-3; none
    # on the other three workloads -- are not the engine's to match: it tracks the fp64 oracle there, tests/test_hip_parity.py)
    assert len(eng_out - ref_out) <= max(1, int(0.005 * A)) and np.median(d) < 4e-5


# ------------------------------------------------------------------ single-scene chains on geometry records (k_attn_chain<1, 4, 3, GEO>)
@pytest.mark.parametrize("case", ["cfg2", "cfg1", "cfg3_b2", "isolated", "replay"])
def test_geometry_record_chains_against_the_operand_image_chains_and_the_oracle(case):
    """Up to 256 rows the default implementation runs the fused chains of the encoder's a2a layers, the generator and the policy on
    k_attn_chain<1, 4, 3, GEO>: k_chain16's edge arithmetic on 32-byte geometry records behind fp32 GEMV node stages (ps_attn.h,
    c16_lat_* in ps_chain16.h).  ps_set_chain_impl(1) keeps the operand-image edge phase (k_relpe_tiles) it replaced: two summation
    orders of one layer.  Replan 0 of both within 1e-4 of the fp64 oracle and 2e-5 of each other; closed loop within the band (or the
    scene's own fp32 floor on the small chaotic specs); rows without any edge (an agent alone in its scene) take the empty-softmax
    path of both; 257 rows and more stay on the operand-image build (two workgroups per CU)."""
    import torch
    from prosim_amd.engine import Engine
    from prosim_amd.spec import DEMO_SPEC
    from oracle import prosim_oracle as orc
    if case.startswith("cfg"):
        spec = DEMO_SPEC
        cfg = int(case[3])
        scene = synth.baseline_scene(spec, cfg, seed=0, batch=2 if case.endswith("b2") else (1 if cfg == 2 else None))
    elif case == "isolated":   # one or two agents per scene, far apart: rows with no a2a / p2p / a2p edge at all
        spec = SMALL_SPEC
        scene = synth.make_scene(spec, 2, 40, batch=3, seed=5, goal=True, ragged=True, square=400.0)
    else:                       # log-replay agents leave and enter: candidate filters and dead rows in the searches that feed the records
        spec = SMALL_SPEC
        scene = synth.make_scene(spec, 24, 60, batch=2, seed=6, tags=True, replay=0.4, enter=0.5)
    w = weights.init_weights(spec, 0)
    out = {}
    eng = Engine(spec, w)
    try:
        for impl in (0, 1):
            eng.set_chain_impl(impl)
            eng.set_scene(scene)
            eng.rollout()
            pol = eng.policy_rows
            out[impl] = (eng.get("motion_pred")[:, pol].copy(), eng.padded("traj").copy())
            rows = eng.num_agents
            A = int(pol.sum())
    finally:
        eng.close()
    assert rows <= 256
    pm = scene["prompt_mask"].astype(bool)   # (log-replay agents are rows of the engine but no policy agents: compared through the masks)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
        o32 = orc.rollout(w, spec, scene) if spec is SMALL_SPEC else None
    ref_mp = o64["motion_pred"][:A].numpy()
    ref_traj = o64["traj"].numpy()
    for impl in (0, 1):
        assert np.isfinite(out[impl][1][pm]).all()
        assert np.abs(out[impl][0][0] - ref_mp).max() < 1e-4, (case, impl)
    assert np.abs(out[0][0][0] - out[1][0][0]).max() < 2e-5, case   # replan 0: two summation orders of the same layers
    err = np.abs(out[0][1] - ref_traj)[pm].max()
    if o32 is None:
        assert err < 1e-4, (case, err)            # BASELINE-size scenes: the band itself (cfg3 seed 0: all 256 agents, tests/golden/known_cut_agents.json)
    else:
        floor = float(np.abs(o32["traj"].numpy() - ref_traj)[pm].max())
        assert err < 3 * floor + 1e-4, (case, err, floor)


# ------------------------------------------------------------------ one launch per radius search (k_radius_geo)
@pytest.mark.parametrize("case", ["cfg2", "cfg3_b2_rows16", "small_conditions", "isolated", "replay", "cap_hit"])
def test_one_launch_search_returns_the_bits_of_the_three_launch_search(case):
    """ps_set_search_impl(0) (default): a radius search whose edges feed a geometry-record chain is ONE launch -- one scan per query, the CSR
    prefix from counts the workgroups publish to each other, esrc / edst and the 32-byte records written by the search's own waves
    (k_radius_geo, ps_chain16.h).  ps_set_search_impl(1): the count / fill / k_edge_geo launches of rounds 1-4.  Same edges in the same
    order and the same record arithmetic: the closed-loop results must be EQUAL bit for bit -- with self matches to drop (p2p), candidate
    filters (log-replay agents), rows without any edge, a neighbour cap that every query hits, in latency and in throughput mode, and
    over REPLAYS of the captured graph (the kernel clears its own flags: a stale flag would show up as a different CSR).  The wait of the
    look-back is bounded: a count that is not published in time is recomputed by the waiting wave (ps_set_search_impl(2) forces that path
    for every count), so no workgroup depends on another one's progress."""
    from prosim_amd.engine import Engine
    from prosim_amd.spec import DEMO_SPEC
    import dataclasses
    rows = 0
    if case == "cfg2":
        spec, scene = DEMO_SPEC, synth.baseline_scene(DEMO_SPEC, 2, seed=3, batch=1)
    elif case == "cfg3_b2_rows16":
        spec, scene, rows = DEMO_SPEC, synth.baseline_scene(DEMO_SPEC, 3, seed=1, batch=2), 16
    elif case == "small_conditions":
        spec, scene, rows = SMALL_SPEC, synth.make_scene(SMALL_SPEC, 24, 160, batch=3, seed=5, goal=True, tags=True, ragged=True), 16
    elif case == "isolated":
        spec, scene = SMALL_SPEC, synth.make_scene(SMALL_SPEC, 2, 40, batch=3, seed=5, goal=True, ragged=True, square=400.0)
    elif case == "replay":
        spec, scene = SMALL_SPEC, synth.make_scene(SMALL_SPEC, 24, 60, batch=2, seed=6, tags=True, replay=0.4, enter=0.5)
    else:   # every policy / generator query runs into its cap: the list is cut in index order
        spec = dataclasses.replace(SMALL_SPEC, pol_max_neigh=5, dec_max_neigh=7)
        scene = synth.make_scene(spec, 20, 90, batch=2, seed=9, goal=True)
    w = weights.init_weights(spec, 0)
    eng = Engine(spec, w)
    try:
        out, nodes = {}, {}
        for impl in (1, 0, 2):   # 2: the look-back never waits -- every count of a lower query is recomputed by the waiting wave (the fallback path, forced)
            eng.set_search_impl(impl)
            eng.set_chain_rows(rows)
            eng.set_scene(scene)
            got = []
            for _ in range(3):   # a capture and two replays
                eng.rollout()
                got.append((eng.padded("traj").copy(), eng.get("motion_pred").copy(), eng.padded("vel").copy()))
            for g in got[1:]:
                for a, b in zip(g, got[0]):
                    assert np.array_equal(a, b), (case, impl, "replay differs from capture")
            out[impl], nodes[impl] = got[0], eng.graph_nodes
    finally:
        eng.close()
    for impl in (0, 2):
        for a, b in zip(out[impl], out[1]):
            assert np.array_equal(a, b), (case, impl)
    assert np.isfinite(out[0][0][scene["prompt_mask"].astype(bool)]).all()
    assert nodes[0] < nodes[1], nodes   # two launches less per search
