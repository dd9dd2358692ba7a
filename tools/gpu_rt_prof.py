"""One row-tile PointNet configuration, many launches (for rocprofv3 --pmc passes): PS_RT_WHICH / PS_RT_N / PS_RT_P / PS_RT_MT."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from prosim_amd import weights
from prosim_amd.spec import DEMO_SPEC
from prosim_amd.engine import Engine
spec = DEMO_SPEC
eng = Engine(spec, weights.init_weights(spec, 0))
which, n, P, mt = (int(os.environ.get(k, d)) for k, d in (("PS_RT_WHICH", 1), ("PS_RT_N", 128), ("PS_RT_P", 11), ("PS_RT_MT", 1)))
rng = np.random.RandomState(0)
x = rng.randn(n, P, spec.map_dim if which == 0 else spec.obs_dim).astype(np.float32)
m = rng.rand(n, P) > 0.1
y, ms = eng.test_pointnet_mt(which, x, m, mt, iters=10)
print(f"which={which} n={n} P={P} mt={mt}: {ms*1e3:.1f} us")
eng.close()
