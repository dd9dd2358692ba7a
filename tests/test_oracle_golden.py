"""CPU tests: the oracle (oracle/prosim_oracle.py) against the committed golden fixtures that
tests/gen_golden.py produced from the reference's own Python."""
import os

import numpy as np
import pytest
import torch

from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC
from oracle import prosim_oracle as orc
from golden_cases import FULL_CASES, SPECS, digest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pure():
    return np.load(os.path.join(GOLD, "ref_pure_primitives.npz"))


@pytest.fixture(scope="module")
def Wt():
    return orc.W(weights.init_weights(DEMO_SPEC, 0))


def test_weight_init_is_stable(pure):
    assert digest(weights.init_weights(DEMO_SPEC, 0)) == str(pure["weight_digest"])
    assert weights.n_params(DEMO_SPEC) == 11399220  # every learnable tensor of the reference's demo model
    assert weights.n_params(DEMO_SPEC.replace(drag_mlp_layers=0)) == 11316148   # without the drag-point PointNet


@pytest.mark.parametrize("tag,in_dim,pre,n,prefix", [("map", 11, 3, 5, "scene_encoder.map_encoder"),
                                                     ("obs", 24, 1, 3, "scene_encoder.obs_encoder")])
def test_pointnet_ref_pure(pure, Wt, tag, in_dim, pre, n, prefix):
    x, mk = torch.from_numpy(pure[f"pointnet_{tag}_x"]), torch.from_numpy(pure[f"pointnet_{tag}_mask"])
    y = orc.pointnet(Wt, prefix, in_dim, 128, pre, n, x, mk)
    assert np.abs(y.numpy() - pure[f"pointnet_{tag}_y"]).max() < 1e-6
    assert np.all(y.numpy()[0, 3] == 0)  # all-invalid polyline encodes to 0 (pointnet_encoder.py:56-59)


def test_fourier_ref_pure(pure):
    y = orc.fourier_fix(torch.from_numpy(pure["fourier_x"]), 32.0)
    assert np.array_equal(y.numpy(), pure["fourier_y"])
    assert np.array_equal(orc.fourier_div(32.0).numpy(), pure["fourier_div32"])


def test_geometry_ref_pure(pure):
    assert np.array_equal(orc.wrap_angle(torch.from_numpy(pure["wrap_x"])).numpy(), pure["wrap_y"])
    w = pure["wrap_y"]
    assert w.min() >= -np.pi - 1e-6 and w.max() < np.pi + 1e-6
    assert np.array_equal(orc.rel_traj_coord_to_last_step(torch.from_numpy(pure["reltraj_x"])).numpy(), pure["reltraj_y"])
    tr, vl = torch.from_numpy(pure["reltraj_x"]), torch.from_numpy(pure["relvel_x"])
    th = torch.atan2(tr[..., 2], tr[..., 3])[..., -1:]
    assert np.array_equal(orc.batch_rotate_2d(vl, -th).numpy(), pure["relvel_y"])


def test_head_ref_pure(pure, Wt):
    y, _ = orc.cg_stacked(Wt, "policy.act_decoder.CG_decode", torch.from_numpy(pure["cg_anchor"]), torch.from_numpy(pure["cg_ctx"]))
    assert np.abs(y.numpy() - pure["cg_y"]).max() < 1e-6
    z = orc.mlp(Wt, "policy.act_decoder.motion_head", [128, 128, 64, 50], y, True, False)
    assert np.abs(z.numpy() - pure["motion_head_y"]).max() < 1e-5


@pytest.mark.parametrize("name", list(FULL_CASES))
def test_rollout_ref_standins(name):
    """Full ProSim.forward(batch,'val') fixtures (reference Python + stand-ins).  Closed-loop fp32
    noise grows ~1.5-2x per replan, so later replans are held to a multiple of the fp32 floor the
    generator measured against the fp64 restatement; replan 0 is held to 1e-4 absolute."""
    sname, kw, wseed = FULL_CASES[name]
    spec = SPECS[sname]
    g = np.load(os.path.join(GOLD, f"ref_standins_{name}.npz"))
    w = weights.init_weights(spec, wseed)
    scene = synth.make_scene(spec, **kw)
    assert digest(scene) == str(g["scene_digest"]) and digest(w) == str(g["weight_digest"])
    if "mode_choice" in g.files:      # TOP_K > 1: the reference's own mode draws, replayed
        scene["mode_choice"] = g["mode_choice"]
    if "action_noise" in g.files:     # RANDOM_NOISE_STD > 0: the reference's own noise draws, replayed
        scene["action_noise"] = g["action_noise"]
    with torch.no_grad():
        o = orc.rollout(w, spec, scene)
    A = int(scene["prompt_mask"].sum())
    assert np.abs(o["motion_pred"][:A].numpy() - g["motion_pred"][:A]).max() < 1e-4
    if spec.use_goal_pred_loss:
        assert np.abs(o["reconst_pred"].numpy() - g["reconst_pred"]).max() < 1e-5
    else:   # (no pred_mlp: neither the reference nor the oracle returns it)
        assert "reconst_pred" not in o and g["reconst_pred"].shape[0] == 0
    floor = dict(zip(("traj", "vel", "motion_pred"), g["fp32_floor"]))
    for k in ("traj", "vel", "motion_pred"):
        err = np.abs(o[k].numpy() - g[k]).max()
        assert err < 3 * floor[k] + 1e-4, (k, err, floor[k])


def test_neighbour_semantics():
    """radius: strict <, index-order truncation; knn: (distance, index) order, self first."""
    pos = torch.tensor([[0., 0.], [3., 0.], [0., 4.], [5., 0.], [0., 0.]])
    b = torch.zeros(5, dtype=torch.long)
    yi, xi = orc.radius_edges(pos, b, pos[:1], b[:1], 5.0, 10)
    assert xi.tolist() == [0, 1, 2, 4]          # |(5,0)| == r is excluded
    yi, xi = orc.radius_edges(pos, b, pos[:1], b[:1], 5.0, 2)
    assert xi.tolist() == [0, 1]                # first two in index order
    yi, xi = orc.radius_edges(pos, b, pos, b, 5.0, 2, drop_self=True)
    assert xi[yi == 0].tolist() == [1, 2]       # cap+1 = 3 kept (0,1,2), then self removed
    yi, xi = orc.knn_edges(pos, b, pos[:1], b[:1], 3)
    assert xi.tolist() == [0, 4, 1]             # tie at d=0 -> lower index first
    b2 = torch.tensor([0, 0, 1, 1, 1])
    yi, xi = orc.knn_edges(pos, b2, pos, b2, 4)
    assert sorted(xi[yi == 0].tolist()) == [0, 1] and sorted(xi[yi == 3].tolist()) == [2, 3, 4]


def test_pair_metric_oracle_vs_reference_fixture():
    """oracle/metric_oracle.py against tests/golden/ref_pair_metric.npz, which the reference's own PairMotionPred
    (metrics/motion_pred.py:111-199 over loss/loss_func.py:215-313) produced from the same seeded inputs: the five logged
    scalars per batch, the accumulated values after two updates, and the chained trajectories bit for bit."""
    import torch
    from golden_cases import make_pair_metric_inputs, digest
    from oracle import metric_oracle as mo
    g = np.load(os.path.join(GOLD, "ref_pair_metric.npz"))
    keys = ("ade", "fde", "min_ade", "min_fde", "rollout_ade")
    sums = {k: [0.0, 0.0] for k in keys}
    for i, seed in enumerate((0, 1)):
        d = make_pair_metric_inputs(seed)
        assert digest(d) == str(g[f"digest_{i}"])
        o = mo.pair_motion_pred(torch.from_numpy(d["motion_pred"]), torch.from_numpy(d["motion_prob"]), torch.from_numpy(d["tgt"]),
                                torch.from_numpy(d["mask"]), d["bidx"], d["tidx"], d["nidx"], 10)
        for j, k in enumerate(keys):
            assert abs(float(o[k]) - g[f"single_{i}"][j]) < 1e-6 * max(1.0, abs(g[f"single_{i}"][j])), (i, k)
        assert np.array_equal(o["tgt_rollout"].numpy(), g[f"tgt_rollout_{i}"])
        assert np.array_equal(o["pred_rollout"].numpy(), g[f"pred_rollout_{i}"])
        # MeanMetric over updates: pair metrics pool their finite values, rollout_ade averages the per-update scalars
        for k, vec in (("ade", o["pair_ade"]), ("fde", o["pair_fde"]), ("min_ade", o["pair_min_ade"]), ("min_fde", o["pair_min_fde"])):
            ok = ~vec.isnan()
            sums[k][0] += float(vec[ok].double().sum())
            sums[k][1] += int(ok.sum())
        sums["rollout_ade"][0] += float(o["rollout_ade"])
        sums["rollout_ade"][1] += 1
        for j, k in enumerate(keys):
            want = g["after_updates"][i][j]
            assert abs(sums[k][0] / sums[k][1] - want) < 1e-6 * max(1.0, abs(want)), (i, k, sums[k][0] / sums[k][1], want)
