import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import mrc_oracle as O
    from mrchip import optimiser
    h, w, n, c = map(int, sys.argv[1:5])
    rng = np.random.RandomState(1)
    m = rng.rand(h, w) < 0.3
    img = rng.randint(0, 256, (h, w) if c == 1 else (h, w, 3)).astype(np.uint8)
    if c == 1:
        got = optimiser.optimise_gray2(m, img, w, h, n); exp = O.optimise_gray2(m, img, w, h, n)
    else:
        got = optimiser.optimise_rgb2(m, img, w, h, n); exp = O.optimise_rgb2(m, img, w, h, n)
    bad = np.argwhere(got != exp)
    print(h, w, n, c, 'mismatch', len(bad), bad[:6].tolist())
    sys.exit(0)
for (h, w, n, c) in [(33, 47, 10, 3), (33, 47, 3, 3), (33, 47, 10, 1), (33, 46, 10, 3), (33, 45, 10, 3), (33, 63, 10, 3), (20, 517, 3, 3), (20, 517, 3, 1), (20, 517, 10, 3), (20, 513, 3, 3), (20, 527, 3, 3), (20, 1000, 3, 3), (20, 1111, 3, 3), (389, 517, 3, 3)]:
    r = subprocess.run([sys.executable, __file__, str(h), str(w), str(n), str(c)], capture_output=True, text=True)
    out = (r.stdout + r.stderr).strip().splitlines()
    print(out[0] if r.returncode == 0 and out else ('%d %d %d %d FAULT rc=%d: %s' % (h, w, n, c, r.returncode, [l for l in out if 'fault' in l.lower()][:1])), flush=True)
