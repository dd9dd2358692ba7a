import numpy as np
import pytest

from prosim_amd import synth
from prosim_amd.postprocess import replicate_scene
from prosim_amd.spec import SMALL_SPEC


def test_replicate_scene_shapes():
    s = synth.make_scene(SMALL_SPEC, 5, 7, batch=1, seed=0, goal=True)
    s["fut_obs_input"] = np.zeros((SMALL_SPEC.n_replans - 1, 1, 5, 11, 24), np.float32)
    r = replicate_scene(s, 4)
    assert r["obs_input"].shape[0] == 4 and r["cond"]["goal"]["input"].shape == (4, 5, 3)
    assert r["fut_obs_input"].shape[:2] == (SMALL_SPEC.n_replans - 1, 4)
    assert np.array_equal(r["map_pos"][3], s["map_pos"][0])


def test_replicate_scene_keys_the_frame_axis_on_the_field_name():
    """A 2-replan spec has ONE fut_obs frame: [1, 1, N, ...] must still be tiled on axis 1 (the batch), not axis 0."""
    spec = SMALL_SPEC.replace(max_steps=20)
    s = synth.make_scene(spec, 5, 7, batch=1, seed=0, replay=0.4)
    assert s["fut_obs_input"].shape[:2] == (1, 1)
    r = replicate_scene(s, 3)
    for k in ("fut_obs_input", "fut_obs_mask", "fut_obs_pos", "fut_obs_head"):
        assert r[k].shape[:2] == (1, 3), k
    with pytest.raises(ValueError):
        replicate_scene(synth.make_scene(spec, 5, 7, batch=2, seed=0), 3)
