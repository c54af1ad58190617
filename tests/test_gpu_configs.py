"""GPU parity at the batch configurations of BASELINE.json, against per-page digests of the REFERENCE
(tests/golden/configs.json, made by make_golden.py `configs`):

  configs[2]  batch of 64 pages 3300x4600, window 51: gray Sauvola-only batch and RGB full decomposition
  configs[3]  512-page 4000x3000 RGB stack with per-page hOCR, page i -> rank i mod G for G in {1,2,4,8};
              the shards run one after the other on this GPU, every page's three outputs digest-checked
"""
import numpy as np
import pytest

from mrchip import _lib, mrc, synth
from mrchip.dist import shard_pages
from helpers import load_configs, sha, sha_many

pytestmark = pytest.mark.gpu

C2_DISTINCT = 64


def _c2_seed(i):
    """page i of the 512-page stack: the 64 reference-digested pages in a stride that mixes the seeds over ranks"""
    return 202 + (i * 5) % C2_DISTINCT


@pytest.fixture(scope='module')
def c2_pages():
    cfg = load_configs()['c2_pages']
    seeds = sorted(int(s) for s in cfg)
    assert len(seeds) == C2_DISTINCT
    made = synth.synth_pages([dict(w=4000, h=3000, channels=3, seed=s, noise_sigma=6.0, line_div=60) for s in seeds])
    pages = dict(zip(seeds, made))
    for s, h in zip(seeds, sha_many([pages[s][0] for s in seeds])):
        assert h == cfg[str(s)]['in'], 'synthetic page %d differs from the one the reference digested' % s
    return pages, cfg


def test_config3_gray_sauvola_batch_of_64():
    cfg = load_configs()['c3_gray_pages']
    seeds = sorted(int(s) for s in cfg)
    made = synth.synth_pages([dict(w=3300, h=4600, channels=1, seed=s, noise_sigma=6.0) for s in seeds])
    imgs = {s: m[0] for s, m in zip(seeds, made)}
    for s in seeds:
        assert sha(imgs[s]) == cfg[str(s)]['in']
    ctx = _lib.default_context()
    bt = mrc.Batch(ctx, 64, 3300, 4600, 1)
    order = [seeds[(i * 3) % len(seeds)] for i in range(64)]
    for i, s in enumerate(order):
        bt.upload(i, imgs[s])
    bt.threshold(None, 0.34)                                  # one Sauvola launch over the 64 pages
    masks = [bt.download_mask(i) for i in range(64)]
    for i, (s, h) in enumerate(zip(order, sha_many(masks))):
        assert h == cfg[str(s)]['out'], (i, s)
        assert int(masks[i].sum()) == cfg[str(s)]['sum']
    bt.close()


def test_config3_rgb_full_batch_of_64():
    cfg = load_configs()['c3_rgb_pages']
    seeds = sorted(int(s) for s in cfg)
    made = synth.synth_pages([dict(w=3300, h=4600, channels=3, seed=s, noise_sigma=6.0, line_div=60) for s in seeds])
    pages = dict(zip(seeds, made))
    for s in seeds:
        assert sha(pages[s][0]) == cfg[str(s)]['in']
    order = [seeds[(i * 3 + i // 8) % len(seeds)] for i in range(64)]
    n = 0
    stream = mrc.decompose_stream(((pages[s][0], pages[s][1]) for s in order), bg_downsample=3, batch_pages=64)
    for s, (m, fg, bg) in zip(order, stream):
        want = cfg[str(s)]
        hm, hf, hb = sha_many([m, fg, bg], 3)
        assert (hm, hf, hb) == (want['mask'], want['fg'], want['bg']), (n, s)
        assert list(bg.shape) == want['bg_shape']
        n += 1
    assert n == 64


@pytest.mark.parametrize('world', [8, 4, 2, 1])
def test_config4_512_page_stack_sharded(c2_pages, world):
    pages, cfg = c2_pages
    total = 512
    seen = 0
    for rank in range(world):
        mine = shard_pages(total, rank, world)                     # page i -> rank i mod world
        stream = mrc.decompose_stream(((pages[_c2_seed(i)][0], pages[_c2_seed(i)][1]) for i in mine), bg_downsample=3,
                                      batch_pages=32, mask_format='bool')
        batch = []
        for i, (m, fg, bg) in zip(mine, stream):
            batch.append((i, m, fg, bg))          # views stay valid for batch_pages (32) further pages
            if len(batch) == 16 or i == mine[-1]:
                hs = sha_many([a for rec in batch for a in rec[1:]])
                for k, rec in enumerate(batch):
                    want = cfg[str(_c2_seed(rec[0]))]
                    assert tuple(hs[3 * k:3 * k + 3]) == (want['mask'], want['fg'], want['bg']), (world, rank, rec[0])
                    assert int(rec[1].sum()) == want['mask_sum']
                seen += len(batch)
                batch = []
    assert seen == total
