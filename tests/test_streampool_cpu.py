"""StreamPool bookkeeping without a GPU (one idle 8-page slot per distinct page size does not scale to books
whose pages all differ): slots are stubbed, the eviction / sizing rules are the real ones."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'archive-pdf-tools_amd'))


class _Slot:
    live = 0
    closed = []

    def __init__(self, ctx, n, w, h, c):
        self.key = (n, w, h, c)
        self.stamp = 0
        _Slot.live += 1

    def close(self):
        _Slot.live -= 1
        _Slot.closed.append(self.key)


def _pool(monkeypatch, max_bytes):
    from mrchip import mrc
    monkeypatch.setattr(mrc, '_StreamSlot', _Slot)
    _Slot.live, _Slot.closed = 0, []
    return mrc, mrc.StreamPool(ctx=object(), max_bytes=max_bytes)


def test_many_distinct_page_sizes_stay_under_the_budget(monkeypatch):
    mrc, pool = _pool(monkeypatch, max_bytes=3 * mrc_bytes(1, 4000, 3000, 3))
    for i in range(300):                       # a book in which every page has its own size, 4 batches in flight
        sl = pool.take(1, 4000 + i, 3000, 3)
        assert sl.key == (1, 4000 + i, 3000, 3)          # sized to the run, not to batch_pages
        pool.give(sl)
        assert pool.bytes_held() <= pool.max_bytes
        assert _Slot.live <= 3
    assert len(_Slot.closed) >= 297
    pool.close()
    assert _Slot.live == 0


def mrc_bytes(n, w, h, c):
    from mrchip import mrc
    return mrc._slot_bytes(n, w, h, c)


def test_same_geometry_is_reused_and_kept(monkeypatch):
    mrc, pool = _pool(monkeypatch, max_bytes=10 * mrc_bytes(8, 4000, 3000, 3))
    a = pool.take(8, 4000, 3000, 3, capacity=8)
    b = pool.take(8, 4000, 3000, 3, capacity=8)
    pool.give(a)
    c = pool.take(3, 4000, 3000, 3)            # the short last batch of the run fits the idle 8-page slot
    assert c is a
    pool.give(b); pool.give(c)
    other = pool.take(1, 1000, 1000, 1)        # far below the budget: nothing is closed
    pool.give(other)
    assert _Slot.closed == [] and _Slot.live == 3
    pool.close()


def test_eviction_prefers_other_geometries_and_least_recently_used(monkeypatch):
    one = mrc_bytes(1, 2000, 2000, 3)
    mrc, pool = _pool(monkeypatch, max_bytes=3 * mrc_bytes(1, 2000, 2003, 3))
    s1 = pool.take(1, 2000, 2000, 3); pool.give(s1)
    s2 = pool.take(1, 2000, 2001, 3); pool.give(s2)
    s3 = pool.take(1, 2000, 2002, 3); pool.give(s3)
    s4 = pool.take(1, 2000, 2002, 3)           # same geometry as s3: s3 is taken, nothing new
    assert s4 is s3
    s5 = pool.take(1, 2000, 2002, 3)           # a second one of that geometry: the oldest OTHER idle slot goes
    assert _Slot.closed == [(1, 2000, 2000, 3)]
    in_flight = pool.take(1, 2000, 2003, 3)    # s2 goes; slots in flight (s3, s5) are never touched
    assert _Slot.closed == [(1, 2000, 2000, 3), (1, 2000, 2001, 3)]
    for s in (s4, s5, in_flight):
        pool.give(s)
    pool.close()
    assert _Slot.live == 0


def test_stage_clock_books_gpu_time_per_key_and_the_rest_on_the_phase_key():
    """timing_data (mrc.py:363-468): every key = the GPU time of its kernels (differences of the profile's running totals),
    the phase's remainder on the key that stands for the phase, a caller's profiling session left as it was."""
    from mrchip import mrc

    class Ctx:
        def __init__(self, on):
            self.prof_on = on
            self.calls = []
            self.t = {'luma601': 1.0, 'sauvola_boxes': 2.0, 'hocr_commit': 0.5, 'dwt_dd_f32': 0.25, 'optimise_rgb': 8.0}

        def prof_enable(self, on):
            self.calls.append(on)
            self.prof_on = on

        def prof_report(self):
            return {k: {'ms': v, 'launches': 1, 'alg_bytes': 0.0} for k, v in self.t.items()}

    ctx = Ctx(False)
    clock = mrc._StageClock(ctx, True)
    assert ctx.calls == [True]
    ctx.t['luma601'] += 2.0; ctx.t['sauvola_boxes'] += 3.0; ctx.t['dwt_dd_f32'] += 1.0; ctx.t['median_hist'] = 0.5
    g = clock.lap()
    assert abs(g['grey_conversion'] - 0.002) < 1e-12 and abs(g['hocr_mask_gen'] - 0.003) < 1e-12 and abs(g['est_1'] - 0.0015) < 1e-12
    td = []
    mrc._book(td, 0.010, 'hocr_mask_gen', [('grey_conversion', g['grey_conversion']), ('hocr_mask_gen', g['hocr_mask_gen']),
                                           ('est_1', g['est_1'])])
    assert [k for k, _ in td] == ['grey_conversion', 'hocr_mask_gen', 'est_1']
    assert abs(sum(v for _, v in td) - 0.010) < 1e-12 and abs(dict(td)['hocr_mask_gen'] - (0.003 + 0.0035)) < 1e-12
    assert clock.lap()['partial_blur'] == 0.0                      # nothing ran since the last lap
    clock.close()
    assert ctx.calls == [True, False]
    ctx2 = Ctx(True)                                               # the caller is profiling: not switched, not switched off
    c2 = mrc._StageClock(ctx2, True); c2.close()
    assert ctx2.calls == [] and ctx2.prof_on
    c3 = mrc._StageClock(Ctx(False), False)                        # no timing_data: nothing happens
    assert c3.lap() == {} and c3.ctx.calls == []
