import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import mrc_oracle as O
from mrchip import optimiser, _lib
ctx = _lib.default_context(); ctx.prof_enable(True)
for (h, w, n, c) in [(50, 700, 3, 3), (50, 700, 10, 3), (50, 4000, 10, 3), (50, 4096, 10, 3), (50, 700, 10, 1)]:
    rng = np.random.RandomState(1)
    m = rng.rand(h, w) < 0.3
    img = rng.randint(0, 256, (h, w) if c == 1 else (h, w, 3)).astype(np.uint8)
    got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(m, img, w, h, n)
    exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(m, img, w, h, n)
    bad = np.argwhere(got != exp)
    print(h, w, n, c, 'mismatch', len(bad), bad[:5].tolist(), flush=True)
print(ctx.prof_report())
