"""TRAJ.PRED_MODE 'cluster' / 'mlp' on the host side (CPU): the folded cluster anchors against the oracle's module math, the tensors
the engine is handed, checkpoint loading with the goal-cluster file, and the head's output width."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import weights  # noqa: E402
from prosim_amd.spec import SMALL_SPEC  # noqa: E402
from oracle import prosim_oracle as orc  # noqa: E402


def test_cluster_anchors_fold_equals_the_module_math():
    spec = SMALL_SPEC.replace(k_pred_mode="cluster", motion_k=3)
    w = weights.init_weights(spec, 0)
    Wt = {k: torch.from_numpy(v).double() for k, v in w.items()}
    pe = orc.fourier_fix(Wt[weights.CLUSTER_GOALS], spec.hidden // 2)
    ref = orc.mlp(Wt, "policy.act_decoder.cluster_mlp", [spec.hidden, spec.hidden], pe, False, False).numpy()
    a = weights.cluster_anchors(spec, w)
    assert a.shape == (3, spec.hidden) and a.dtype == np.float32
    assert np.abs(a - ref).max() < 2e-5 and np.abs(ref).max() > 0.1      # fp32 sin / cos of arguments up to ~2 pi * 60
    et = weights.engine_tensors(spec, w)
    t = et["policy.act_decoder.motion_anchors.weight"]
    assert t.shape == (3 * spec.num_agent_types, spec.hidden)
    for ty in range(spec.num_agent_types):                                  # the same K rows for every agent type
        assert np.array_equal(t[3 * ty:3 * ty + 3], a)
    assert weights.CLUSTER_GOALS not in et and not any(".cluster_mlp." in k for k in et)
    assert weights.engine_tensors(SMALL_SPEC, weights.init_weights(SMALL_SPEC, 0)) is not None


def test_checkpoint_loading_needs_the_goal_cluster_file():
    spec = SMALL_SPEC.replace(k_pred_mode="cluster", motion_k=2)
    w = weights.init_weights(spec, 1)
    sd = weights.to_reference_state_dict(spec, w)
    assert weights.CLUSTER_GOALS not in sd                                  # (a file next to the checkpoint, not a parameter)
    with pytest.raises(KeyError):
        weights.from_state_dict(spec, sd)
    w2 = weights.from_state_dict(spec, sd, k_goals=w[weights.CLUSTER_GOALS])
    assert set(w2) == set(w) and all(np.array_equal(w[k], w2[k]) for k in w)


def test_head_shapes_by_mode():
    d = SMALL_SPEC.hidden
    for mode, k, out, cg, anchors in (("anchor", 3, 50, True, True), ("cluster", 3, 50, True, False), ("mlp", 2, 100, False, False)):
        spec = SMALL_SPEC.replace(k_pred_mode=mode, motion_k=k)
        sh = weights.param_shapes(spec)
        assert sh["policy.act_decoder.motion_head.mlp.6.weight"] == (out, d // 2) and spec.head_out_dim == out
        assert ("policy.act_decoder.CG_decode.CGs.0.MLP.0.weight" in sh) == cg
        assert ("policy.act_decoder.motion_anchors.weight" in sh) == anchors
    with pytest.raises(ValueError):
        SMALL_SPEC.replace(k_pred_mode="vel_pred")


def test_fewer_pe_frequency_bands_are_zero_padded_exactly():
    """*.ATTN.PE_NUM_FREQ below the engine's 64 bands: weights.engine_tensors hands over the 64-band embedding whose extra bands have
    frequency 0 and zero first-Linear columns -- the oracle's FourierEmbedding on the padded tensors equals the original's."""
    spec = SMALL_SPEC.replace(pol_learnable_pe=True, pe_num_freq=16)
    w = weights.init_weights(spec, 0)
    et = weights.engine_tensors(spec, w)
    p = "policy.act_decoder.a2p_rel_pe_emb"
    assert et[p + ".freqs.weight"].shape == (3, 64) and et[p + ".mlps.1.0.weight"].shape == (spec.hidden, 129)
    assert w[p + ".freqs.weight"].shape == (3, 16)                          # the caller's dict is untouched
    x = torch.tensor(np.random.RandomState(0).uniform(-3, 3, (50, 3)), dtype=torch.float64)
    a = orc.fourier_learn({k: torch.from_numpy(v).double() for k, v in w.items()}, p, x)
    b = orc.fourier_learn({k: torch.from_numpy(np.asarray(v)).double() for k, v in et.items()}, p, x)
    assert float((a - b).abs().max()) < 1e-12
    with pytest.raises(ValueError):
        weights.engine_tensors(spec.replace(pe_num_freq=128), weights.init_weights(spec.replace(pe_num_freq=128), 0))
