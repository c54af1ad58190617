"""Listing checks that need no GPU: hipcc cross-compiles the two kernels with hand-counted asynchronous loads to
assembly, tools/isa_inflight.py scans them for reads of a register between the asm load that targets it and the asm
`s_waitcnt vmcnt(N)` covering it ("the torn copy"), and the listing must show no spills in those kernels
(a spill of an in-flight register stores stale data)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'archive-pdf-tools_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _listing(tmp_path, name):
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    out = str(tmp_path / (name + '.s'))
    r = subprocess.run([HIPCC, '-std=c++17', '-O3', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-S',
                        '--cuda-device-only', os.path.join(CSRC, name + '.hip'), '-o', out], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _scan(listing, key):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_inflight.py'), listing, key], capture_output=True, text=True)
    return r.returncode, r.stdout


def _scratch_of(listing, key):
    txt = open(listing).read()
    m = re.search(r'\.amdhsa_kernel (\S*%s\S*)(.*?)\.end_amdhsa_kernel' % re.escape(key), txt, re.S)
    assert m, key
    return int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2)).group(1))


def test_sauvola_row_queues_are_never_read_in_flight(tmp_path):
    lst = _listing(tmp_path, 'k_sauvola')
    # the table kernels of pages (8 columns, 16 waves) and hOCR boxes (4 columns, two polarities), the fp64 kernels
    # (the last template argument of the table kernel: stores per row counted into the vmcnt waits -- 0 = the compiler's stores)
    for key in ('sauvola_tab_kernelILi8ELb1ELb0ELi8ELi16ELi4ELi3E', 'sauvola_tab_kernelILi8ELb0ELb0ELi8ELi16ELi4ELi3E',       # 3: the bit rows only
                'sauvola_tab_kernelILi8ELb1ELb0ELi8ELi16ELi4ELi2E', 'sauvola_tab_kernelILi8ELb1ELb0ELi8ELi16ELi4ELi1E',
                'sauvola_tab_kernelILi8ELb1ELb0ELi8ELi16ELi4ELi0E', 'sauvola_tab_kernelILi8ELb0ELb0ELi8ELi16ELi4ELi2E',
                'sauvola_tab_kernelILi4ELb1ELb1ELi8ELi8', 'sauvola_tab_kernelILi4ELb1ELb1ELi32ELi8',
                'sauvola_tab_kernelILi8ELb0ELb0ELi8ELi16ELi4ELi0E', 'sauvola_kernelILi8ELb1ELb0ELi8E', 'sauvola_kernelILi4ELb1ELb1ELi32E',
                'sauvola_kernelILi16ELb1ELb0ELi32E'):
        rc, out = _scan(lst, key)
        assert rc == 0, (key, out[-1500:])
        assert _scratch_of(lst, key) == 0, key
    # every Sauvola kernel of the library is free of scratch (the 8-column two-polarity table kernel
    # spilled 8 VGPRs; box launches now stay on 4 columns and that instantiation is gone), and the wide two-polarity
    # fp64 kernels that page-sized boxes with wide windows DO reach are scanned like the others
    import re as _re
    txt = open(lst).read()
    names = _re.findall(r'\.amdhsa_kernel (\S*sauvola\S*)', txt)
    assert len(names) > 20
    assert not [n for n in names if 'sauvola_tab_kernelILi8ELb0ELb1' in n or 'sauvola_tab_kernelILi8ELb1ELb1' in n]
    for n in names:
        assert _scratch_of(lst, n) == 0, n
    for key in ('sauvola_kernelILi8ELb1ELb1ELi32E', 'sauvola_kernelILi8ELb0ELb1ELi32E', 'sauvola_kernelILi16ELb1ELb1ELi32E',
                'sauvola_kernelILi16ELb0ELb1ELi32E'):
        rc, out = _scan(lst, key)
        assert rc == 0, (key, out[-1500:])


def test_the_page_layer_optimise_kernel_spills_nothing_inside_its_row_loops(tmp_path):
    """optimise_band_kernel<3, 2, 1024, true> -- THE kernel of configs[1] -- carried 26 spilled VGPRs,
    some reloaded behind loop labels.  Page-layer launches now take the instance with three row loops instead of six
    (PAGES): a few launch-lifetime values are spilled at entry and reloaded once per queue entry (loop depth 1 = the
    walker's loop over bands); no scratch access inside a row loop (depth >= 2)."""
    lst = _listing(tmp_path, 'k_optimise')
    lines = open(lst).read().split('\n')
    key = 'optimise_band_kernelILi3ELi2ELi1024ELb1ELb1E'
    assert _scratch_of(lst, key) <= 32
    i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0])
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
    label, deep = '', []
    for i in range(i0, i1):
        t = lines[i].strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            label = t
        elif t.startswith('scratch_'):
            m = re.search(r'Depth=(\d+)', label)
            if m and int(m.group(1)) >= 2:
                deep.append((i - i0, label[:60], t[:50]))
    assert not deep, deep[:5]


def test_the_float32_gaussian_sees_its_stores_acknowledged_before_the_in_kernel_clean_up(tmp_path):
    """gauss_fast_kernel<R <= 3> rewrites, after its second barrier, single bytes of dwords that other waves of the workgroup
    stored before it.  hipcc's workgroup-scope barrier waits for lgkmcnt only, so the kernel carries its own
    `s_waitcnt vmcnt(0)` in front of that barrier (k_gauss.hip): the listing must show it, directly before the last
    s_barrier of the kernel; the radii with a separate fix launch (R >= 4) need none."""
    lst = _listing(tmp_path, 'k_gauss')
    lines = open(lst).read().split('\n')
    for R, wanted in ((1, True), (2, True), (3, True), (4, False), (8, False)):
        key = 'gauss_fast_kernelILi%dE' % R
        i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0])
        i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
        body = [l.strip() for l in lines[i0:i1] if l.strip() and not l.strip().startswith((';', '.'))]
        bars = [i for i, t in enumerate(body) if t.startswith('s_barrier')]
        assert len(bars) >= 2, (key, len(bars))
        before = body[max(0, bars[-1] - 4):bars[-1]]
        has = any(t.startswith('s_waitcnt') and 'vmcnt(0)' in t for t in before)
        assert has == wanted, (key, before)
