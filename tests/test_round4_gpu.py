"""Round-4 GPU tests: MODEL.REL_POS_EDGE_FUNC 'knn' beyond its reference-made fixture (log-replay / entering agents, the stateless
policy call, both fused-chain kernels), the row-tile kernels against the staged ones (PointNet by row tiles per wave, whole
rollouts by ps_set_row_impl), the standalone geometry-record edge kernel of the split s2s path."""
import numpy as np
import pytest
import torch

from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from oracle import prosim_oracle as orc
from golden_cases import SPECS

pytestmark = pytest.mark.gpu
TOL = 1e-4


def err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


@pytest.mark.parametrize("impl", [1, 2])
def test_knn_edge_sets_with_log_replay_and_entering_agents(impl):
    """REL_POS_EDGE_FUNC 'knn' (sym_coord.py:85-96, act_decoder.py:249-261) where the candidate sets are FILTERED: only policy agents
    are prompts (knn_graph over a subset), log-replay agents drop out of / enter the scene between replans.  Against the fp64
    oracle: replan 0 to 1e-4, the closed loop to the scene's fp32 floor.  impl 1 = k_attn_chain (rel-PE
    operand images over the knn CSR), 2 = k_chain16 (geometry records)."""
    from prosim_amd.engine import Engine
    spec = SPECS["small_knn"]
    w = weights.init_weights(spec, 0)
    scene = synth.make_scene(spec, 24, 90, batch=3, seed=41, goal=True, tags=True, ragged=True, replay=0.4, enter=0.5)
    with torch.no_grad():
        o64 = orc.rollout(w, spec, scene, dtype=torch.float64)
        o32 = orc.rollout(w, spec, scene)
    floor = float((o32["traj"].double() - o64["traj"]).abs().max())
    eng = Engine(spec, w)
    try:
        eng.set_chain_impl(impl)
        eng.set_scene(scene)
        eng.rollout()
        # (the engine searches for every observed agent row, the oracle for the policy agents only: edge totals are not comparable here)
        pol = eng.policy_rows
        A = int(pol.sum())
        mp = eng.get("motion_pred")[:, pol]
        assert err(mp[0], o64["motion_pred"][:A].numpy()) < TOL
        pm = scene["prompt_mask"].astype(bool)
        assert err(eng.padded("traj")[pm], o64["traj"].numpy()[pm]) < 3 * floor + TOL
    finally:
        eng.close()


def test_row_tile_pointnet_equals_the_staged_kernel_for_every_tiling():
    """k_pointnet_rt (ps_rowtile.h) by row tiles per wave against k_pointnet_mfma on ragged masks, map- and history-shaped inputs:
    two fp32 evaluation orders of the same encoder (pointnet_encoder.py:24-62)."""
    from prosim_amd.engine import Engine
    spec = DEMO_SPEC
    eng = Engine(spec, weights.init_weights(spec, 0))
    rng = np.random.RandomState(3)
    try:
        for which, n, P, C in ((0, 301, 19, spec.map_dim), (1, 77, 11, spec.obs_dim), (0, 5, 32, spec.map_dim), (1, 40, 3, spec.obs_dim)):
            x = rng.randn(n, P, C).astype(np.float32)
            m = rng.rand(n, P) > 0.3
            m[0] = False                      # a polyline without a valid point keeps a zero feature
            ref, _ = eng.test_pointnet_mt(which, x, m, -1)
            assert (ref[0] == 0).all()
            for mt in (0, 1, 2, 3, 4, 5):
                if mt and 16 * mt < P:
                    continue
                builds = {1: 16, 2: 16, 3: 8, 4: 8, 5: 4}   # widest lane count of the builds with mt tiles per wave (kRtShapes)
                if mt and mt * builds[mt] < P:   # (round 5: a forced tiling without a build for this P is an error, not a silent fall-back)
                    with pytest.raises(RuntimeError, match="no row-tile build"):
                        eng.test_pointnet_mt(which, x, m, mt)
                    continue
                y, _ = eng.test_pointnet_mt(which, x, m, mt)
                assert (y[0] == 0).all()
                assert err(y, ref) < 2e-6, (which, n, P, mt, err(y, ref))
    finally:
        eng.close()


@pytest.mark.parametrize("rows", [0, 16])
def test_row_impls_agree_on_the_benchmark_batch(rows):
    """ps_set_row_impl 0 (row-tile kernels, k_edge_rows behind them in the split path) against 1 (the staged kernels of rounds 1-3) on
    the 8-scene benchmark batch: scene tokens to 2e-5, the closed loop to 1e-4; and 2 (the 16-row workgroup edge kernel k_edge16) /
    11..13 (forced row tiles per wave in the node halves) against 0 BIT FOR BIT -- a row's result depends neither on the tiling of the
    node GEMMs nor on which form of the edge phase walked its edges."""
    from prosim_amd.engine import Engine
    spec = DEMO_SPEC
    w = weights.init_weights(spec, 0)
    parts = [synth.baseline_scene(spec, 2, seed=i, batch=1) for i in range(8)]
    scene = {k: (np.concatenate([p[k] for p in parts]) if not isinstance(parts[0][k], dict) else
                 {ck: {f: np.concatenate([p[k][ck][f] for p in parts]) for f in parts[0][k][ck]} for ck in parts[0][k]}) for k in parts[0]}
    eng = Engine(spec, w)
    try:
        ref = exact = None
        for impl in (1, 0, 2, 11, 12, 13):
            try:
                eng.set_row_impl(impl)
            except RuntimeError as ex:       # (2 and 13 exist in -DPS_EXPERIMENTS builds of the library only: round 5)
                assert impl in (2, 13) and "experiments build" in str(ex), (impl, str(ex))
                continue
            eng.set_chain_impl(2)            # (k_chain16 with the SPLIT s2s path in either mode: the node halves + edge kernel under test)
            eng.set_chain_rows(rows)
            eng.set_scene(scene)
            eng.rollout()
            tok, traj = eng.get("scene_tokens"), eng.padded("traj")
            assert np.isfinite(traj).all()
            if ref is None:
                ref = (tok, traj)
            assert err(tok, ref[0]) < 2e-5 and err(traj, ref[1]) < TOL, (impl, err(tok, ref[0]), err(traj, ref[1]))
            if impl == 0:
                exact = (tok, traj)
            elif impl != 1:
                assert np.array_equal(tok, exact[0]) and np.array_equal(traj, exact[1]), (impl, err(tok, exact[0]), err(traj, exact[1]))
    finally:
        eng.close()
