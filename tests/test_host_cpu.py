"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/prosim_hip.h
declares (no compute calls), weight container round-trips, synthetic generators are stable."""
import ctypes
import os
import re

import numpy as np
import pytest

from prosim_amd import synth, weights
from prosim_amd.spec import DEMO_SPEC, SMALL_SPEC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "prosim_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from prosim_amd import engine
    lib = ctypes.CDLL(engine.lib_path())
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in prosim_hip.h but not exported"
    assert set(engine.EXPORTS) == set(syms)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from prosim_amd.engine import Engine
    with pytest.raises(RuntimeError, match="libprosim_hip error"):
        Engine(SMALL_SPEC, weights.init_weights(SMALL_SPEC, 0))


def test_bad_config_and_missing_weight_are_rejected():
    from prosim_amd import engine
    lib = engine.load_library()
    cfg = engine.PsConfig(hidden=64, heads=8, head_dim=16)
    h = ctypes.c_void_p()
    rc = lib.ps_create(ctypes.byref(cfg), 0, None, None, None, ctypes.byref(h))
    assert rc == -1 and b"hidden=128" in lib.ps_last_error()


def test_mlp_scene_encoders_are_refused_loudly():
    """MODEL.SCENE_ENCODER.MAP_TYPE / OBS_TYPE 'mlp' (scene_encoder/map_encoder.py:5, obs_encoder.py:19, picked at base.py:20-21) are legal
    keys of the reference's registry with no engine counterpart: ps_create says so before it touches the GPU (round 6)."""
    from prosim_amd import engine
    from prosim_amd.engine import Engine
    with pytest.raises(ValueError, match="'pointnet' or 'mlp'"):
        SMALL_SPEC.replace(map_encoder_type="cnn")
    for kw, key in ((dict(map_encoder_type="mlp"), "MAP_TYPE"), (dict(obs_encoder_type="mlp"), "OBS_TYPE")):
        with pytest.raises(RuntimeError, match=f"{key} 'mlp'.*not built"):
            Engine(SMALL_SPEC.replace(**kw), weights.init_weights(SMALL_SPEC, 0))
    lib = engine.load_library()
    cfg = engine.PsConfig(hidden=128, heads=8, head_dim=16, motion_k=1, state_dim=5, target_steps=10, map_encoder_mlp=1)
    h = ctypes.c_void_p()
    assert lib.ps_create(ctypes.byref(cfg), 0, None, None, None, ctypes.byref(h)) == -1 and b"MAP_TYPE 'mlp'" in lib.ps_last_error()


def test_weight_container_matches_reference_naming():
    shapes = weights.param_shapes(DEMO_SPEC)
    assert shapes["scene_encoder.a2a_attn_layers.0.to_g.weight"] == (128, 256)
    assert shapes["policy.act_decoder.motion_head.mlp.6.weight"] == (50, 64)
    assert shapes["scene_encoder.map_encoder.pre_mlps.mlp.6.weight"] == (128, 128)
    assert "scene_encoder.a2a_attn_layers.0.attn_prenorm_x_dst.weight" not in shapes      # alias, not a parameter
    assert "decoder.s2p_attn_layers.0.attn_prenorm_x_dst.weight" in shapes                # bipartite: own LN
    w = weights.init_weights(SMALL_SPEC, 3)
    sd = weights.to_reference_state_dict(SMALL_SPEC, w)
    assert "scene_encoder.a2a_attn_layers.0.attn_prenorm_x_dst.weight" in sd
    w2 = weights.from_state_dict(SMALL_SPEC, sd)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    with pytest.raises(KeyError):
        weights.from_state_dict(SMALL_SPEC, {})


def test_synth_layouts():
    s = synth.make_scene(DEMO_SPEC, 8, 16, batch=2, seed=0, goal=True, tags=True, ragged=True)
    assert s["map_input"].shape == (2, 16, 19, 11) and s["obs_input"].shape == (2, 8, 11, 24)
    assert not np.isnan(s["obs_input"][s["obs_mask"]]).any()                  # InputMaskData invariant (format_utils.py:43)
    assert np.isnan(s["obs_input"][~s["obs_mask"]]).all()
    assert s["cond"]["goal"]["input"].shape == (2, 8, 3)
    for i in range(5):
        synth.baseline_scene(DEMO_SPEC, i) if i != 3 else synth.baseline_scene(DEMO_SPEC, i, batch=1)


def test_gpu_tests_do_not_import_the_reference_harness():
    """The -m gpu modules (and the data module they share) read fixture CASES from tests/golden_cases.py; tests/gen_golden.py -- the generator,
    which imports oracle/ref_harness.py and through it /root/reference -- is for the build container only."""
    import glob
    tests = os.path.join(ROOT, "tests")
    for path in glob.glob(os.path.join(tests, "test_*gpu*.py")) + [os.path.join(tests, n) for n in ("test_hip_parity.py", "golden_cases.py", "oracle_cache.py", "parity_table.py")]:
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+gen_golden\b", src, flags=re.M), path
        assert "ref_harness" not in src, path
