"""The fp64-oracle cache behind the -m gpu parity gates (tests/oracle_cache.py): every committed file carries the digests of
its scene, its weights, its ModelSpec and of the oracle's own source, all of them current (a change of the oracle, the spec or
the weight layout without `python tests/gen_golden.py oracle_cache` fails HERE, on the CPU, instead of letting the GPU gates
pass against outdated expectations -- ADVICE round 3); and one small key is recomputed from scratch and compared."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_cache as oc          # noqa: E402
import gen_golden as gg            # noqa: E402
from oracle import prosim_oracle as orc   # noqa: E402


def test_every_cached_rollout_is_current():
    code = oc.oracle_code_digest()
    seen = set()
    for key, spec, w, scene, collect, floor, slim in gg.oracle_cache_workloads():
        g = np.load(os.path.join(oc.CACHE, key + ".npz"))
        assert str(g["scene_digest"]) == oc._digest({k: v for k, v in scene.items() if not k.startswith("_")}), key
        assert str(g["weight_digest"]) == oc._digest(w), key
        assert str(g["spec_digest"]) == oc.spec_digest(spec), key
        assert str(g["oracle_code_digest"]) == code, f"{key}: written by another version of the oracle (run tests/gen_golden.py oracle_cache)"
        seen.add(key + ".npz")
    assert seen == set(os.listdir(oc.CACHE))


def test_spec_digest_sees_the_caps():
    from prosim_amd.spec import DEMO_SPEC
    assert oc.spec_digest(DEMO_SPEC) != oc.spec_digest(DEMO_SPEC.replace(dec_max_neigh=DEMO_SPEC.dec_max_neigh + 1))


def test_a_small_cached_key_recomputes_to_the_cached_values():
    wl = {k: (spec, w, scene) for k, spec, w, scene, *_ in gg.oracle_cache_workloads()}
    spec, w, scene = wl["no_truncation_cfg1"]
    cached = oc.oracle64("no_truncation_cfg1", spec, w, scene)
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    with torch.no_grad():
        fresh = orc.rollout(w, spec, scene, dtype=torch.float64)
    # (stored as float32 of the float64 result: coordinates of tens of metres, 4e-6 of rounding)
    assert np.abs(fresh["traj"].numpy() - cached["traj"].numpy()).max() < 1e-5
    assert fresh["edges"] == cached["edges"]
