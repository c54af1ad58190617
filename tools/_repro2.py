import sys
sys.path.insert(0, 'archive-pdf-tools_amd'); sys.path.insert(0, 'oracle')
import numpy as np
import mrc_oracle as O
from mrchip import optimiser
rng = np.random.RandomState(5)
for (h, w, c, n) in [(11, 4613, 3, 14), (11, 4613, 3, 10), (30, 4613, 3, 14), (11, 5000, 3, 14), (11, 4613, 1, 14), (5, 8200, 3, 14)]:
    for dens in (0.05, 0.5, 0.95):
        img = rng.randint(0, 256, (h, w, c) if c == 3 else (h, w)).astype(np.uint8)
        mask = (rng.rand(h, w) < dens).astype(np.uint8)
        print('case', h, w, c, n, dens, flush=True)
        got = (optimiser.optimise_gray2 if c == 1 else optimiser.optimise_rgb2)(mask, img, w, h, n)
        exp = (O.optimise_gray2 if c == 1 else O.optimise_rgb2)(mask, img, w, h, n)
        assert np.array_equal(got, exp)
print('ok')
