import sys, time
sys.path.insert(0, 'archive-pdf-tools_amd'); sys.path.insert(0, 'oracle')
import numpy as np
import mrc_oracle as O
from mrchip import optimiser, _lib
rng = np.random.RandomState(3)
ctx = _lib.default_context()
for (h, w, c, n, dens) in [(300, 1300, 3, 3, 0.1), (300, 1300, 3, 10, 0.9), (200, 4000, 3, 10, 0.92), (200, 4000, 3, 3, 0.08), (120, 8000, 3, 10, 0.9),
                           (120, 8000, 1, 3, 0.1), (77, 1001, 1, 10, 0.5), (64, 700, 3, 11, 0.5), (50, 5003, 3, 7, 0.3), (3000, 4000, 3, 10, 0.92), (3000, 4000, 3, 3, 0.08)]:
    img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
    mask = (rng.rand(h, w) < dens).astype(np.uint8)
    f = optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2
    got = f(mask, img, w, h, n)
    t0 = time.time(); got = f(mask, img, w, h, n); dt = time.time() - t0
    exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
    bad = np.argwhere(got != exp)
    print((h, w, c, n, dens), 'ok' if len(bad) == 0 else ('MISMATCH %d first %s cols %s' % (len(bad), bad[:3].tolist(), sorted(set((bad[:, 1] // 4 * 4).tolist()))[:12])), '%.1f ms' % (dt * 1e3), flush=True)
