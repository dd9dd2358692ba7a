import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prosim_amd import weights
from prosim_amd.spec import SMALL_SPEC
from prosim_amd.engine import Engine
eng = Engine(SMALL_SPEC, weights.init_weights(SMALL_SPEC, 0))
for mb in (1, 12):
    for nwg in (1, 16, 128, 256):
        for depth in (1, 2, 3):
            ms = eng.test_stream(mb, nwg, depth)
            print(f"{mb:3d} MB x {nwg:3d} WG depth {depth}: {ms*1e3:8.1f} us  -> {mb*1.048576/ms:7.1f} GB/s per WG, {nwg*mb*1.048576/ms/1e3:7.2f} TB/s total")
