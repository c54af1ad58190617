"""Listing checks that need no GPU: hipcc cross-compiles the kernels with hand-counted asynchronous loads to assembly,
tools/isa_vmflow.py follows every asm load to a wait that covers it along every path of the kernel's control-flow graph
("the torn copy"; round 6: path-aware, every instantiation), and the listing must show no spills in those
kernels (a spill of an in-flight register stores stale data)."""
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


def _vmflow(listing, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_vmflow.py'), listing] + list(args), capture_output=True, text=True)
    return r.returncode, r.stdout + r.stderr


def _scratch_of(listing, key):
    txt = open(listing).read()
    m = re.search(r'\.amdhsa_kernel (\S*%s\S*)(.*?)\.end_amdhsa_kernel' % re.escape(key), txt, re.S)
    assert m, key
    return int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2)).group(1))


def test_sauvola_kernels_have_no_scratch(tmp_path):
    """A spill of a slot register between its asm load and its wait would store stale data: every Sauvola kernel of the
    library is free of scratch (the 8-column two-polarity table kernel spilled 8 VGPRs; box launches now
    stay on 4 columns and that instantiation is gone).  The in-flight reads themselves are the next test's business
    (tools/isa_vmflow.py, path-aware; the linear scan tools/isa_inflight.py it replaces reports out-of-line blocks of the
    round-6 loop shape that no path reaches with a load in flight)."""
    lst = _listing(tmp_path, 'k_sauvola')
    txt = open(lst).read()
    names = re.findall(r'\.amdhsa_kernel (\S*sauvola\S*)', txt)
    assert len(names) > 20
    assert not [n for n in names if 'sauvola_tab_kernelILi8ELb0ELb1' in n or 'sauvola_tab_kernelILi8ELb1ELb1' in n]
    for n in names:
        assert _scratch_of(lst, n) == 0, n


def test_every_path_from_an_asm_load_to_its_wait_in_every_sauvola_kernel(tmp_path):
    """the path-aware check.  tools/isa_vmflow.py builds each kernel's control-flow graph from the
    listing and follows every hand-counted asm load to a wait that covers it along EVERY path (tile-edge paths, the row
    loop's early exit, the fp64 / table / per-lane decision branches): no instruction may mention a slot register whose
    load can still be in flight, no load may be in flight at s_endpgm.  All instantiations of the library, not a sample.
    Then the tool itself: three faults planted in the listing must each be reported."""
    lst = _listing(tmp_path, 'k_sauvola')
    txt = open(lst).read()
    names = re.findall(r'\.amdhsa_kernel (\S*sauvola\S*)', txt)
    tiled = [n for n in names if 'sauvola_kernelI' in n or 'sauvola_tab_kernelI' in n]
    assert len(tiled) >= 36, len(tiled)
    rc, out = _vmflow(lst, '--all', 'sauvola')
    assert rc == 0, out[-3000:]
    for n in tiled:          # every one of them was seen with its twelve loads and six waits
        line = next(l for l in out.split('\n') if l.startswith(n + ':'))
        assert '12 asm loads, 6 asm vmcnt waits, 0 findings' in line, line
    # --- the checker is not blind: planted faults in the page kernel's listing ---
    key = 'sauvola_tab_kernelILi8ELb1ELb0ELi8ELi16ELi4ELi3E'
    lines = txt.split('\n')
    i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
    asm_waits = [i for i in range(i0, i1) if lines[i].strip().startswith('s_waitcnt vmcnt(') and lines[i - 1].strip().startswith(';;#ASMSTART')]
    counted = [i for i in asm_waits if 'vmcnt(0)' not in lines[i]]
    assert len(counted) == 4 and len(asm_waits) == 6, (len(counted), len(asm_waits))

    def mutated(edit):
        m = list(lines)
        edit(m)
        p = str(tmp_path / 'mut.s')
        open(p, 'w').write('\n'.join(m))
        return _vmflow(p, key)

    def weaker(m):           # one wait of the row loop allows one more operation in flight than the program counted
        n = int(re.search(r'vmcnt\((\d+)\)', m[counted[0]]).group(1))
        m[counted[0]] = m[counted[0]].replace('vmcnt(%d)' % n, 'vmcnt(%d)' % (n + 1))
    rc, out = mutated(weaker)
    assert rc == 1 and 'may be in flight' in out, out[-800:]

    def no_closing_wait(m):  # the tile's closing vmcnt(0) gone: the queue's last loads are in flight at s_endpgm
        m[asm_waits[-1]] = '\ts_nop 0'
    rc, out = mutated(no_closing_wait)
    assert rc == 1 and 'still in flight' in out, out[-800:]

    def early_copy(m):       # a copy of a slot register right behind its load (what a spill or a phi copy would be)
        i = next(i for i in range(i0, i1) if lines[i].strip().startswith('global_load_dwordx2') and lines[i - 1].strip().startswith(';;#ASMSTART'))
        reg = re.search(r'v\[(\d+):', lines[i]).group(1)
        m.insert(i + 2, '\tv_mov_b32_e32 v255, v%s' % reg)
    rc, out = mutated(early_copy)
    assert rc == 1 and 'v_mov_b32_e32 v255' in out, out[-800:]


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
    # the strip schedule's hand-off loads (asm loads with vmcnt(0) waits): every path, every instantiation
    rc, out = _vmflow(lst, '--all', 'optimise_strip_kernel')
    assert rc == 0 and out.count('0 findings') >= 8, out[-2000:]


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
