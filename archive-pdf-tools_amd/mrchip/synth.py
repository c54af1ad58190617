"""Deterministic, RNG-free synthetic scanned pages + hOCR line boxes.

Input generator for tests and bench.py (SURVEY.md 8d "Synthetic inputs"): every
value is an integer hash (splitmix64 finaliser) of (seed, y, x), so the build
container (golden vectors, numpy 1.26 / 2.2) and the GPU box produce identical
bytes without depending on numpy's Generator stream.

A page is paper (225,218,200) with a slow horizontal gradient, per-pixel
triangular noise of a chosen sigma, and text lines made of dark vertical strokes;
one hOCR line box per text line (+4 px margin) grouped into paragraphs, with a
few low-confidence / blank-text lines (skipped by the reference,
mrc.py:198-203), one light-on-dark line (inverted-polarity branch,
mrc.py:258-261), one pair of overlapping boxes (ordered overwrite, mrc.py:266)
and isolated specks for the mask denoiser.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = 0x9E3779B97F4A7C15
_MASK = (1 << 64) - 1


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _h(*ints):
    """scalar hash of a few python ints -> python int (64 bit)."""
    z = 0
    for v in ints:
        z = (z + (v & _MASK) * _GOLD + 0x632BE59BD9B4E019) & _MASK
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        z ^= z >> 31
    return z


def pixel_hash(seed, h, w):
    """uint64 [h,w] hash field."""
    with np.errstate(over='ignore'):
        ys = (np.arange(h, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93))[:, None]
        xs = (np.arange(w, dtype=np.uint64) * np.uint64(0xCA5A826395121157))[None, :]
        z = ys + xs + np.uint64((seed * _GOLD + 0x2545F4914F6CDD1D) & _MASK)
        return _mix(z)


def synth_page(w, h, channels=3, seed=0, noise_sigma=6.0, line_div=60, specks=True,
               conf_skips=True, inverted_line=True, overlap=True):
    """Returns (img uint8 [h,w(,3)], hocr_word_data) -- hocr_word_data has the
    structure the reference consumes (mrc.py:194-199)."""
    hz = pixel_hash(seed, h, w)
    noise_mul = int(round(noise_sigma * 4096 / 104.5))
    base = (225, 218, 200) if channels == 3 else (218,)
    grad = ((np.arange(w, dtype=np.int32) * 24) // max(1, w - 1)) - 12          # -12..12
    planes = []
    for c in range(channels):
        b0 = ((hz >> np.uint64(16 * c)) & np.uint64(0xFF)).astype(np.int32)
        b1 = ((hz >> np.uint64(16 * c + 8)) & np.uint64(0xFF)).astype(np.int32)
        n = ((b0 + b1 - 255) * noise_mul) >> 12
        planes.append(base[c] + grad[None, :] + n)
    # text lines
    lh = max(6, h // line_div)                  # line height
    pitch = 2 * lh
    top = max(lh, h // 12)
    left, right = w // 10, w - w // 10
    hocr = []
    para = None
    para_left = 0
    li = 0
    y = top
    ink_mask = np.zeros((h, w), dtype=bool)
    inv_band = np.zeros((h, w), dtype=bool)
    ink_val = np.zeros((h, w), dtype=np.int32)
    prev_box = None
    while y + lh + lh // 2 < h - top // 2:
        if para is None or para_left == 0:
            para = {'lines': []}
            hocr.append(para)
            para_left = 3 + _h(seed, li, 1) % 6          # 3..8 lines
            if li:
                y += lh                                    # paragraph gap
                if y + lh + lh // 2 >= h - top // 2:
                    break
        r = _h(seed, li, 2)
        x0 = left + (r % max(1, w // 40))
        x1 = right - ((r >> 16) % max(1, w // 6))
        inverted = inverted_line and li == 5
        # strokes
        x = x0
        k = 0
        words = []
        wx0 = x
        while x < x1:
            rr = _h(seed, li, 3, k)
            sw = 2 + rr % 5                                # stroke width 2..6
            gap = 3 + (rr >> 8) % 7
            if (rr >> 16) % 7 == 0:                        # word gap
                words.append((wx0, x))
                gap += lh
                wx0 = x + sw + gap
            ink = 40 + ((rr >> 24) % 21) - 10
            sh = lh - ((rr >> 32) % max(1, lh // 3))       # stroke height varies
            xe = min(x + sw, x1)
            ink_mask[y + lh - sh:y + lh, x:xe] = True
            ink_val[y + lh - sh:y + lh, x:xe] = ink
            x = xe + gap
            k += 1
        if wx0 < x1:
            words.append((wx0, x1))
        box = [max(0, x0 - 4), max(0, y - 4), min(w, x1 + 4), min(h, y + lh + 4)]
        if inverted:
            inv_band[box[1]:box[3], box[0]:box[2]] = True
        if overlap and li == 8 and prev_box is not None:
            box[1] = prev_box[3] - lh // 2                  # overlaps the previous line's box
        conf = 90
        text = 'w'
        if conf_skips and li % 11 == 7:
            conf = 10                                       # skipped: conf < 20
        if conf_skips and li % 13 == 9:
            text = ' '                                      # skipped: blank text
        para['lines'].append({
            'bbox': [float(b) for b in box],
            'words': [{'text': text, 'confidence': conf if i % 2 == 0 else min(100, conf + 5),
                       'bbox': [float(a), float(box[1]), float(b), float(box[3])]}
                      for i, (a, b) in enumerate(words or [(x0, x1)])],
        })
        prev_box = box
        para_left -= 1
        li += 1
        y += pitch
    out = []
    for c in range(channels):
        p = planes[c]
        # light-on-dark line: dark band, light strokes
        p = np.where(inv_band, 45 + (p - base[c]), p)
        txt = np.where(inv_band, 255 - ink_val - 25, ink_val)
        p = np.where(ink_mask, txt + ((p - base[c]) >> 1), p)
        out.append(p)
    if specks:
        # isolated dark specks (1-3 px) outside of text, ~1 per 4000 px
        sp = ((hz >> np.uint64(48)) % np.uint64(4001)) == np.uint64(0)
        sp2 = np.zeros_like(sp)
        sp2[:, 1:] = sp[:, :-1] & (((hz[:, :-1] >> np.uint64(60)) & np.uint64(1)) == np.uint64(1))
        sp |= sp2
        for c in range(channels):
            out[c] = np.where(sp & ~ink_mask, 30, out[c])
    img = np.stack([np.clip(p, 0, 255).astype(np.uint8) for p in out], axis=-1)
    if channels == 1:
        img = img[:, :, 0]
    return np.ascontiguousarray(img), hocr


def kat_pattern(w, h, channels=1):
    """SURVEY.md 8c known-answer pattern: v=(3x^2+5y^2+7xy+11x+13y+17c) mod 251;
    bars=(12<=y%40<28) & ((x//6)%3==0) & (W//10<x<W-W//10); px = bars ? v%64 : 160+v%90."""
    y = np.arange(h, dtype=np.int64)[:, None, None]
    x = np.arange(w, dtype=np.int64)[None, :, None]
    c = np.arange(channels, dtype=np.int64)[None, None, :]
    v = (3 * x * x + 5 * y * y + 7 * x * y + 11 * x + 13 * y + 17 * c) % 251
    bars = ((y % 40 >= 12) & (y % 40 < 28)) & ((x // 6) % 3 == 0) & (x > w // 10) & (x < w - w // 10)
    px = np.where(bars, v % 64, 160 + v % 90).astype(np.uint8)
    return np.ascontiguousarray(px[:, :, 0] if channels == 1 else px)


def pil_mode_image(rgb, mode):
    """A PIL image of `mode` made from an RGB array, the way tests/golden/make_golden.py made the mode vectors.
    Mode 'P' is built from an explicit 3-3-2 palette (index = RRRGGGBB of the pixel, Image.putpalette) instead of
    `convert('P')`, whose adaptive quantiser differs between Pillow versions: every other mode goes through
    Pillow's fixed-formula conversions."""
    from PIL import Image
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    if mode != 'P':
        return Image.fromarray(rgb).convert(mode)
    idx = (rgb[..., 0] & 0xE0) | ((rgb[..., 1] & 0xE0) >> 3) | (rgb[..., 2] >> 6)
    pal = []
    for i in range(256):
        r, g, b = (i >> 5) & 7, (i >> 2) & 7, i & 3
        pal += [(r * 255) // 7, (g * 255) // 7, (b * 255) // 3]
    im = Image.fromarray(idx.astype(np.uint8), 'P')
    im.putpalette(pal)
    return im


def synth_pages(specs, threads=None):
    """[synth_page(**spec) for spec in specs] on a thread pool (numpy releases the GIL in the array work);
    spec = dict of synth_page's arguments."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    if threads is None:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
    threads = max(1, min(threads, 32, len(specs)))
    if threads == 1:
        return [synth_page(**sp) for sp in specs]
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda sp: synth_page(**sp), specs))


def two_level_page(w, h):
    """RGB page of 0 / 255 blocks (9 x 7 px checker): the largest variances a window can have."""
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.where((yy // 9 + xx // 7) % 2 == 0, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2))
