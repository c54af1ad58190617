// optimise_gray2 / optimise_rgb2 (reference: cython/optimiser.pyx:153-273, 280-429) for gfx950 -- the wave-strip
// schedule.  OPT-IN (MRCHIP_OPT_WS=1): bit-exact and tested, but slower than the workgroup schedule of k_optimise.hip
// on MI355X (DESIGN.md 5.R3 has the measurements and the reasons).  Semantics and arithmetic as in k_optimise.hip (same
// packed 16-bit column sums, same exact quotient); what differs is who walks the rows.
//
// The causal window of optimise looks LEFT and UP only: pixel (y, x) needs outputs of rows y-n .. y-1, columns
// x-n .. x-1, never of its own row (pyx:213-220, 250-255).  So a page-layer is cut into column strips of ONE WAVE each
// (<= 62 lanes x 4 columns of core, plus halo lanes on both sides), every wave walks its strip top to bottom at its own
// pace, and the only thing that crosses a strip boundary is, per finished row, the n output columns at the boundary:
// per 4 columns three (RGB) / one (gray) 8-byte units {dword of output bytes, tag}, written and read by relaxed
// system-scope atomics (write-through `sc0 sc1`; MI355X_MICROARCH.md, inter-workgroup visibility: a data-tagged unit
// needs no fence and no flag, wherever the two waves run).  There is no workgroup barrier anywhere.
//
//  * halo: lanes left of the core rebuild the vertical FIR sums of the neighbour's last columns from the inputs and get
//    the vertical IIR sums of those columns from the hand-off units (kept n + 1 rows deep in an LDS ring: a row is needed
//    again when it leaves the sums); lanes right of the core need FIR sums only.
//    HL = ceil(n / 4) lanes on the left, HR = ceil((n - 1) / 4) on the right.
//  * the horizontal windows slide over two wave-private LDS rows (entries of the 256 columns of the wave): LDS
//    operations of one wave execute in order, so there is no wait between publish and window reads either.
//  * the 1-bpp mask is loaded ONCE per row (the entering row); the lane's nibble goes into a shift register of
//    2n + 1 nibbles from which the current and the leaving row's selection come out again.
//  * a row's inputs are requested two rows ahead into one of two register sets by hand-counted asm loads (ws_rows).
//  * rows without "rare" pixels in the window are cheap: a dense layer goes dormant (copies rows, rebuilds its sums from
//    the image when needed), a sparse one takes a short path (ws_rows).
//  * dispatch order = (page-layer, strip), left strips first: a consumer is never resident without its producer
//    having been dispatched, so the polls always end; they are bounded all the same and a timeout is reported
//    (OptMail error word, checked by the host at the next synchronisation point).
//
// Algorithmic bytes: (1 + 2C) * w * h per call.
#include <cstdlib>
#include <type_traits>

#include "mrchip_internal.h"

namespace mrchip {

typedef const unsigned __attribute__((address_space(1))) *gc_u32p;
typedef unsigned __attribute__((address_space(1))) *g_u32p;
typedef uint8_t __attribute__((address_space(1))) *g_u8p;

typedef unsigned long long u64;

struct WsLaunch {
    u64 *mail;              // hand-off units {data dword, tag}; a job's part starts at OptJob::ws_mail
    unsigned tagbase;       // epoch << 16
    unsigned *err;          // page-locked host word, set when a poll gave up
    int smax;               // grid = jobs * smax single-wave workgroups
};

// One hand-off unit = 8 bytes {one dword of output bytes, tag}, read and written by ONE relaxed system-scope atomic
// (`global_load/store_dwordx2 sc0 sc1`): single-copy atomic, so a unit whose tag matches carries that row's bytes, and
// -- unlike an inline-asm load -- the compiler knows these loads: a request that is in flight across the loop's back edge
// cannot be copied or spilled before it has landed (a first version used 16-byte asm loads into a loop-carried register
// quad; when the register allocator put a copy between the load and its wait, a copy that straddled the arrival took
// the new tag with the previous row's bytes).
__device__ __forceinline__ u64 ws_mail_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ws_mail_store(u64 *p, unsigned data, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// 16-bit halves added straight out of the packed pairs (SDWA operand selects)
__device__ __forceinline__ unsigned ws_add_w0w0(unsigned a, unsigned b) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned ws_add_w1w1(unsigned a, unsigned b) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned ws_add_dw0(unsigned a, unsigned b) {      // a + (b & 0xffff)
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned ws_add_dw1(unsigned a, unsigned b) {      // a + (b >> 16)
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// a lane's 4 pixels of a row: one register triple (RGB) or one register (gray), loaded by ONE asm statement
template <int C> struct PxV;
template <> struct PxV<3> { typedef unsigned T __attribute__((ext_vector_type(3))); };
template <> struct PxV<1> { typedef unsigned T; };
template <int C>
__device__ __forceinline__ void px_load_async(typename PxV<C>::T &r, unsigned off, const uint8_t *base) {
    if constexpr (C == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
}
template <int C>
__device__ __forceinline__ unsigned px_dword(const typename PxV<C>::T &r, int q) {
    if constexpr (C == 3) return r[q]; else return r;
}

// NCT: n_size as a compile-time constant (3 and 10, the reference's two call sites mrc.py:413/415, 447/449) or -1 for
// a run-time n <= 11.  16-bit lane capacity as in k_optimise.hip: one FIR accumulator up to n = 8 (and FIR + IIR in one
// for n <= 7), two half-window accumulators up to n = 11.
template <int C, int NCT>
__device__ __forceinline__ void ws_rows(const OptJob &J, unsigned *lds, const WsLaunch &L, const int strip) {
    constexpr int P = 4;
    constexpr int EW = (C == 3) ? 2 : 1;          // dwords per LDS entry
    constexpr int ND = C;                          // dwords of pixel bytes per lane-row (P * C / 4)
    constexpr bool SUMROW = (NCT >= 1 && NCT <= 7);
    constexpr int NH = (NCT >= 0 && NCT <= 8) ? 1 : 2;
    constexpr int HWORDS = NCT >= 0 ? (4 * (2 * NCT + 1) + 31) / 32 : 3;      // mask shift register, 2n + 1 nibbles
    typedef typename PxV<C>::T pxv_t;
    const uint8_t *__restrict__ img = J.img;
    uint8_t *out = J.out;
    const unsigned *mbits = J.mbits;
    const int ipitch = J.ipitch, opitch = J.opitch, mwpr = J.mwpr, w = J.w, h = J.h;
    const int n = NCT >= 0 ? NCT : J.n;
    const int HL = (n + 3) >> 2, HR = (n + 2) >> 2;
    const int l = threadIdx.x;
    const int XS = strip * J.ws_clb * 4;          // first core column
    const int XEc = XS + J.ws_clb * 4;            // one past the last core column (may lie beyond w in the last strip)
    const int XE = min(w, XEc);
    const int x0 = XS - 4 * HL + 4 * l;           // the lane's first column (may be < 0 or >= w)
    const bool act = x0 >= XS && x0 < XE;
    const bool lhalo = strip > 0 && l < HL;
    const bool producer = strip + 1 < J.ws_S && x0 >= XEc - 4 * HL && x0 < XEc;
    (void)HR;

    // column the unconditional row loads use: clamped into the image, so that lanes outside (left of column 0, right of
    // the last column group) read valid memory; their bytes are masked where they are used
    const int xl = min(max(x0, 0), max(0, ((w - 1) / P) * P));
    unsigned vo_px = (unsigned)(xl * C);
    unsigned vo_m = (unsigned)((xl >> 5) * 4);
    const unsigned msh = (unsigned)xl & 31u;
    unsigned colm = 0, coln = 0, pxm[ND];          // 0xFF per valid column / bit per valid column / 0xFF per valid pixel byte
#pragma unroll
    for (int b = 0; b < 4; b++)
        if (x0 + b >= 0 && x0 + b < w) { colm |= 0xffu << (8 * b); coln |= 1u << b; }
#pragma unroll
    for (int q = 0; q < ND; q++) {
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { const int cc = x0 + (4 * q + b) / C; if (cc >= 0 && cc < w) m |= 0xffu << (8 * b); }
        pxm[q] = m;
    }
    const unsigned invn = J.invert ? 0xFu : 0u;

    // wave-private LDS rows: entry of column x0 + j (j in [-n, 4 + n)) at lane base + (d + d / 4) * EW, d = j + n
    const int npad = n;
    const int nent = 64 * P + 2 * npad;
    const int nelem = nent + nent / P + 1;
    unsigned *firA = lds, *iirA = lds + (size_t)nelem * EW;
    for (int i = l; i < 2 * nelem * EW; i += 64) lds[i] = 0;
    const int ebase = 5 * l * EW;
    auto eidx = [&](int j) { const int d = j + npad; return ebase + (d + (d >> 2)) * EW; };

    struct Ent { unsigned d[EW]; };
    auto eadd = [](Ent &a, const Ent &b) {
#pragma unroll
        for (int k = 0; k < EW; k++) a.d[k] += b.d[k];
    };
    auto esub = [](Ent &a, const Ent &b) {
#pragma unroll
        for (int k = 0; k < EW; k++) a.d[k] -= b.d[k];
    };
    auto lds_ld = [&](const unsigned *A, int j) {
        Ent e;
        const unsigned *p = A + eidx(j);
        if constexpr (EW == 2) { uint2 v = *reinterpret_cast<const uint2 *>(p); e.d[0] = v.x; e.d[1] = v.y; }
        else e.d[0] = p[0];
        return e;
    };

    // mailbox slots of this strip's two boundaries (unit index = (((boundary * h + row) * HL + lane-in-boundary) * ND + dword)
    // (uniform base of the boundary + the lane's 32-bit unit offset: one register per direction, not a 64-bit pointer)
    u64 *mail_out_b = nullptr;
    const u64 *mail_in_b = nullptr;
    unsigned mo_off = 0, mi_off = 0;
    {
        u64 *base = L.mail + J.ws_mail;
        mo_off = (unsigned)((producer ? (x0 - (XEc - 4 * HL)) >> 2 : 0) * ND);
        mi_off = (unsigned)((lhalo ? l : 0) * ND);
        mail_out_b = base + (size_t)min(strip, max(J.ws_S - 2, 0)) * h * HL * ND;
        mail_in_b = base + (size_t)max(strip - 1, 0) * h * HL * ND;
    }
    const size_t mrow = (size_t)HL * ND;           // units per row of a boundary
    auto mail_out = [&](int row, int q) { return (mail_out_b + (size_t)row * mrow + q) + mo_off; };

    // selection nibble (bit i: column x0 + i selected = mask bit set, optional inversion, inside the image) -> 0xFF bytes
    auto nib_bytes = [&](unsigned nib) { return (__umul24(nib, 0x00204081u) & 0x01010101u) * 255u; };
    auto nib_of_word = [&](unsigned mw) { return ((mw >> msh) ^ invn) & coln; };
    unsigned hist[HWORDS];
#pragma unroll
    for (int k = 0; k < HWORDS; k++) hist[k] = 0;
    auto hist_push = [&](unsigned nib) {
#pragma unroll
        for (int k = HWORDS - 1; k >= 1; k--) hist[k] = __builtin_amdgcn_alignbit(hist[k], hist[k - 1], 28);
        hist[0] = (hist[0] << 4) | nib;
    };
    auto hist_at = [&](int k) -> unsigned {          // nibble pushed k pushes ago
        const int pos = 4 * k;
        if constexpr (NCT >= 0) return (hist[pos >> 5] >> (pos & 31)) & 0xFu;
        else {
            unsigned wd = hist[0];
            if constexpr (HWORDS > 1) { if (pos >= 32) wd = hist[1]; }
            if constexpr (HWORDS > 2) { if (pos >= 64) wd = hist[2]; }
            return (wd >> (pos & 31)) & 0xFu;
        }
    };

    // FIR entries of the 4 columns of a row: masked pixel bytes + selection bit
    auto fir_entries = [&](const unsigned (&px)[ND], const unsigned on, Ent (&e)[P]) {
        const unsigned on01 = on & 0x01010101u;
        if constexpr (C == 3) {
            const unsigned d0 = px[0] & __builtin_amdgcn_perm(0u, on, 0x01000000u);   // [M0 M0 M0 M1]
            const unsigned d1 = px[1] & __builtin_amdgcn_perm(0u, on, 0x02020101u);   // [M1 M1 M2 M2]
            const unsigned d2 = px[2] & __builtin_amdgcn_perm(0u, on, 0x03030302u);   // [M2 M3 M3 M3]
            e[0].d[0] = __builtin_amdgcn_perm(d1, d0, 0x0c010c00u); e[0].d[1] = __builtin_amdgcn_perm(on01, d0, 0x0c040c02u);
            e[1].d[0] = __builtin_amdgcn_perm(d1, d0, 0x0c040c03u); e[1].d[1] = __builtin_amdgcn_perm(on01, d1, 0x0c050c01u);
            e[2].d[0] = __builtin_amdgcn_perm(d2, d1, 0x0c030c02u); e[2].d[1] = __builtin_amdgcn_perm(on01, d2, 0x0c060c00u);
            e[3].d[0] = __builtin_amdgcn_perm(d2, d2, 0x0c020c01u); e[3].d[1] = __builtin_amdgcn_perm(on01, d2, 0x0c070c03u);
        } else {
            const unsigned d0 = px[0] & on;
            e[0].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c040c00u);
            e[1].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c050c01u);
            e[2].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c060c02u);
            e[3].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c070c03u);
        }
    };
    // IIR entries of the 4 columns of an output row (bytes of columns outside the image must be zero)
    auto iir_entries = [&](const unsigned (&o)[ND], Ent (&e)[P]) {
        if constexpr (C == 3) {
            e[0].d[0] = __builtin_amdgcn_perm(o[1], o[0], 0x0c010c00u); e[0].d[1] = __builtin_amdgcn_perm(0u, o[0], 0x0c0c0c02u);
            e[1].d[0] = __builtin_amdgcn_perm(o[1], o[0], 0x0c040c03u); e[1].d[1] = __builtin_amdgcn_perm(0u, o[1], 0x0c0c0c01u);
            e[2].d[0] = __builtin_amdgcn_perm(o[2], o[1], 0x0c030c02u); e[2].d[1] = __builtin_amdgcn_perm(0u, o[2], 0x0c0c0c00u);
            e[3].d[0] = __builtin_amdgcn_perm(o[2], o[2], 0x0c020c01u); e[3].d[1] = __builtin_amdgcn_perm(0u, o[2], 0x0c0c0c03u);
        } else {
            e[0].d[0] = o[0] & 0xffu; e[1].d[0] = (o[0] >> 8) & 0xffu; e[2].d[0] = (o[0] >> 16) & 0xffu; e[3].d[0] = o[0] >> 24;
        }
    };

    Ent firE[P], iirE[P];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int k = 0; k < EW; k++) { firE[i].d[k] = 0; iirE[i].d[k] = 0; }
    unsigned prev[ND];
#pragma unroll
    for (int q = 0; q < ND; q++) prev[q] = 0;

    auto load_px = [&](int yy, unsigned (&px)[ND]) {
        const int yc = min(max(yy, 0), h - 1);
        gc_u32p pi = (gc_u32p)((img + (size_t)yc * ipitch) + vo_px);
#pragma unroll
        for (int q = 0; q < ND; q++) px[q] = pi[q];
    };
    auto load_mw = [&](int yy) {
        const int yc = min(max(yy, 0), h - 1);
        gc_u32p pm = (gc_u32p)((const uint8_t *)(mbits + (size_t)yc * mwpr) + vo_m);
        return pm[0];
    };
    // "Rare" pixels decide how much of a row has to be done: the selected ones of a sparse layer (fg: the ink), the
    // unselected ones of a dense layer (bg: the inverted mask).  rare_rows counts the rows of the current FIR window
    // [y-n, y+n) that hold a rare pixel in this wave's columns (halo included).  While it is zero,
    //   * sparse layer: all FIR sums are zero, the count of every pixel is the known (y-ys)(x-xs), no pixel of the current
    //     row is selected -- the row takes the short path (no FIR terms, reciprocals from registers, no image row);
    //   * dense layer: every pixel of the window is selected, i.e. out = img on all of those rows: the wave goes DORMANT --
    //     it copies rows (and keeps the hand-off going) without maintaining any sum, and when a row with a rare pixel is about
    //     to enter the window it rebuilds the sums from the 2n image rows of the window (out = img there, so FIR and IIR sums
    //     both come from the image) and carries on.  A strip's sums depend on nothing outside its own columns + halo, so
    //     each wave decides this for itself.
    const bool sparse = !J.invert;
    auto rare_any = [&](unsigned nib) { return sparse ? __any(nib != 0u) : __any(nib != coln); };
    int rare_rows = 0;
    auto fir_apply = [&](const unsigned (&px)[ND], unsigned nib, bool plus) {
        if (sparse && !__any(nib != 0u)) return;
        Ent e[P];
        fir_entries(px, nib_bytes(nib), e);
#pragma unroll
        for (int i = 0; i < P; i++) { if (plus) eadd(firE[i], e[i]); else esub(firE[i], e[i]); }
    };

    // FIR rows [0, n-1) enter before the loop (those below the image as empty rows); row y+n-1 enters at step y
    for (int yy = 0; yy < n - 1; yy++) {
        unsigned nib = 0;
        if (yy < h) {
            unsigned px[ND];
            load_px(yy, px);
            nib = nib_of_word(load_mw(yy));
            fir_apply(px, nib, true);
            if (rare_any(nib)) rare_rows++;
        }
        hist_push(nib);
    }

    // The left halo gets the neighbour's output row y-1 as a granule and needs it again n rows later as the row that
    // leaves the IIR sums: it keeps its last n + 1 granules in a small LDS ring (HL lanes x ND dwords per row)
    unsigned *hring = iirA + (size_t)nelem * EW;
    int hslot = 0;                                   // slot of row y-1 (uniform)

    // One store instruction per row whatever the lane: the last column group of a row may reach up to 3 columns past the
    // image -- into the row's padding (every device image has pitch >= row bytes + 64, mrchip_internal.h), with zeros
    // (the output registers are masked to the image).  A byte-wise tail would make the number of memory operations of a
    // row data-dependent, and the compiler's counted waits (vmcnt) fall back to "everything" behind such a branch.
    auto store_row = [&](int yy, const unsigned (&res)[ND]) {
        if (!act) return;
        uint8_t *o = out + (size_t)yy * opitch + (size_t)x0 * C;
#pragma unroll
        for (int q = 0; q < ND; q++) ((g_u32p)o)[q] = res[q];
    };
    // (y - ys) * (x - xs) as a float per column, and -- for the short path of a sparse layer, where it is the whole count
    // -- its reciprocal and the quotient's offset: they only change while the window still grows (rows 0..n)
    float kf[P], rcK[SUMROW ? P : 1];
#pragma unroll
    for (int i = 0; i < P; i++) kf[i] = 0.0f;
    auto set_kf = [&](int yy) {
#pragma unroll
        for (int i = 0; i < P; i++) {
            kf[i] = (float)(min(yy, n) * min(max(x0 + i, 0), n));          // (y - ys) * (x - xs)
            if constexpr (SUMROW) {
                rcK[i] = __builtin_amdgcn_rcpf(__builtin_fmaxf(kf[i], 1.0f));
            }
        }
    };
    set_kf(0);

    // TWO sets of input registers, used alternately (the row loop is unrolled by two): row y consumes one set and requests
    // the inputs of row y + 2 into the same registers as soon as it is done with them -- every request has more than a whole
    // row to land.  One row ahead the kernel ran at the pace of the memory latency (a lone wave: ~1.5 us per row, of which
    // ~0.4 us are arithmetic), and the compiler's own wait counts collapse to "everything" behind this loop's branches and
    // polls.  So the loads and their waits are written by hand, as in k_sauvola.hip: a load is an asm statement whose output
    // the compiler takes for ready, every use sits behind an asm `s_waitcnt vmcnt(N)` that names the registers (which also
    // pins the order).  vmcnt retires in order and EVERY row issues the same sequence of memory operations
    //     [row store] [hand-off stores: 0 or U] [mw, e, lv, ol, U hand-off units] [c]
    // whatever its mode (dormant, short, general), so N = the operations issued after the one needed, counted for a strip
    // without hand-off stores (more operations in flight than counted -- the stores of a producer, a poll, the loads of a
    // wake-up -- only make a wait stricter).  Each request has ONE site per set, the sets are distinct variables of the two
    // unrolled row bodies, and the listing is checked for spills: a copy of a register between its load and its wait would
    // read the old contents.
    constexpr int U = ND;                              // hand-off units per lane and row
    constexpr int WAIT_TOP = 7 + U;                    // after row y-2's requests: c(y-2); store, requests, c of row y-1
    constexpr int WAIT_C = 11 + 2 * U;                 // after c(y-2): row y-1 whole; store and requests of row y
    struct In {
        unsigned mw;
        pxv_t e, lv, ol, c;
        u64 mbp[U];
    };
    In SA, SB;
    auto request_top = [&](In &S, int ys) {            // inputs of row ys but its current image row
        const int ye = min(max(ys + n - 1, 0), h - 1), yl = min(max(ys - n - 1, 0), h - 1);
        // (a strip without a left neighbour asks for row 0 of the job's first boundary every time: its part of the buffer
        // holds at least one row of units even when the job has a single strip)
        const int yg = strip > 0 ? min(max(ys - 1, 0), h - 1) : 0;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(S.mw) : "v"(vo_m), "s"((const uint8_t *)(mbits + (size_t)ye * mwpr)) : "memory");
        px_load_async<C>(S.e, vo_px, img + (size_t)ye * ipitch);
        px_load_async<C>(S.lv, vo_px, img + (size_t)yl * ipitch);
        px_load_async<C>(S.ol, vo_px, out + (size_t)yl * opitch);
        // (every lane asks, lanes outside the left halo for the halo's first unit)
#pragma unroll
        for (int q = 0; q < U; q++)
            asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1" : "=v"(S.mbp[q]) : "v"(mi_off * 8u), "s"(mail_in_b + (size_t)yg * mrow + q) : "memory");
    };
    auto request_c = [&](In &S, int ys) {
        px_load_async<C>(S.c, vo_px, img + (size_t)min(max(ys, 0), h - 1) * ipitch);
    };
    // (`steady` is a compile-time tag: a run-time choice between two wait statements that name the same registers makes
    // the compiler merge two versions of them -- with copies of registers whose loads are still in flight)
    auto wait_top = [&](In &S, auto steady_t) {
        if constexpr (decltype(steady_t)::value) {
            if constexpr (U == 3) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(S.mw), "+v"(S.e), "+v"(S.lv), "+v"(S.ol), "+v"(S.mbp[0]), "+v"(S.mbp[U > 1 ? 1 : 0]) : "n"(WAIT_TOP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(S.mw), "+v"(S.e), "+v"(S.lv), "+v"(S.ol), "+v"(S.mbp[0]) : "n"(WAIT_TOP) : "memory");
            if constexpr (U == 3) asm volatile("" : "+v"(S.mbp[U > 2 ? 2 : 0]) : : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(S.mw), "+v"(S.e), "+v"(S.lv), "+v"(S.ol), "+v"(S.mbp[0]) : : "memory");
            if constexpr (U == 3) asm volatile("" : "+v"(S.mbp[1]), "+v"(S.mbp[U > 2 ? 2 : 0]) : : "memory");
        }
    };
    auto wait_c = [&](In &S, auto steady_t) {
        if constexpr (decltype(steady_t)::value) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(S.c) : "n"(WAIT_C) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(S.c) : : "memory");
    };
#pragma unroll
    for (int q = 0; q < U; q++) { SA.mbp[q] = 0; SB.mbp[q] = 0; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the warm-up's loads are the compiler's: start from zero
    request_top(SA, 0); request_c(SA, 0);
    request_top(SB, 1); request_c(SB, 1);
    bool dormant = false;

    auto do_row = [&](const int y, In &S, auto steady) {
        // ---- this row's set has landed (waited for BEFORE this row's stores go out: behind them the wait would include
        // their write acknowledgements -- a memory round trip of the write-through hand-off store in every row) ----
        wait_top(S, steady);
        unsigned e_px[ND], lv_px[ND], ol[ND];
#pragma unroll
        for (int q = 0; q < ND; q++) { e_px[q] = px_dword<C>(S.e, q); lv_px[q] = px_dword<C>(S.lv, q); ol[q] = px_dword<C>(S.ol, q); }
        const unsigned mw = S.mw;
        if (y >= 1) {
            const unsigned tag_prev = L.tagbase + (unsigned)(y - 1);
            if (strip > 0) {
                // the neighbour's output row y-1 (asked for two rows ago): poll until it is this launch's
                auto stale = [&]() {
                    bool st = false;
#pragma unroll
                    for (int q = 0; q < U; q++) st |= (unsigned)(S.mbp[q] >> 32) != tag_prev;
                    return __any(lhalo && st);
                };
                int spins = 0;
                while (stale()) {
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int q = 0; q < U; q++)
                        asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1" : "=v"(S.mbp[q]) : "v"(mi_off * 8u), "s"(mail_in_b + (size_t)(y - 1) * mrow + q) : "memory");
                    if constexpr (U == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(S.mbp[0]), "+v"(S.mbp[U > 1 ? 1 : 0]), "+v"(S.mbp[U > 2 ? 2 : 0]) : : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(S.mbp[0]) : : "memory");
                    if (++spins > (1 << 22)) { if (l == 0) __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                }
            }
            store_row(y - 1, prev);
            if (producer) {
#pragma unroll
                for (int q = 0; q < U; q++) ws_mail_store(mail_out(y - 1, q), prev[q], tag_prev);
            }
            if (strip > 0) {
                hslot = hslot == n ? 0 : hslot + 1;
                if (lhalo) {
#pragma unroll
                    for (int q = 0; q < ND; q++) prev[q] = (unsigned)S.mbp[q];
                    // row y-n-1 comes back out of the ring (the slot after the one row y-1 goes into), row y-1 goes in
                    const int sr = hslot == n ? 0 : hslot + 1;
                    unsigned *pr = hring + (sr * HL + l) * ND, *pw = hring + (hslot * HL + l) * ND;
#pragma unroll
                    for (int q = 0; q < ND; q++) { ol[q] = pr[q]; }
#pragma unroll
                    for (int q = 0; q < ND; q++) { pw[q] = prev[q]; }
                }
            }
        }
        // ---- the entering row's selection; dense layer: may this row be a plain copy? ----
        const unsigned nib_e = (y + n - 1 < h) ? nib_of_word(mw) : 0u;
        const bool rare_e = (y + n - 1 < h) && rare_any(nib_e);
        bool copyrow = false;
        copyrow = !sparse && rare_rows == 0 && !rare_e;
        if (copyrow) dormant = true;          // DORMANT row: no unselected pixel in rows [y-n, y+n) of this wave's columns -> out = img
        else if (dormant) {
            // ---- WAKE UP: the state this row would have found had the sums been kept -- FIR sums of rows [y-1-n, y-1+n),
            // IIR sums of rows [y-1-n, y-1) (out = img on all of them: no unselected pixel) and the mask history ----
            dormant = false;
#pragma unroll
            for (int i = 0; i < P; i++)
#pragma unroll
                for (int k = 0; k < EW; k++) { firE[i].d[k] = 0; iirE[i].d[k] = 0; }
            int nfir = 0;
            for (int r = max(y - 1 - n, 0); r < min(y - 1 + n, h); r++) {
                unsigned px[ND];
                load_px(r, px);
#pragma unroll
                for (int q = 0; q < ND; q++) px[q] &= pxm[q];
                Ent e[P];
                iir_entries(px, e);                      // = the FIR entries of an all-selected row without their count
#pragma unroll
                for (int i = 0; i < P; i++) eadd(firE[i], e[i]);
                if (r < y - 1) {
#pragma unroll
                    for (int i = 0; i < P; i++) eadd(iirE[i], e[i]);
                }
                nfir++;
            }
#pragma unroll
            for (int i = 0; i < P; i++) firE[i].d[EW - 1] += ((coln >> i) & 1u) * (unsigned)nfir << 16;
            // history as it stands before this row's push: position k <-> row y+n-2-k, all valid columns selected
#pragma unroll
            for (int k = 0; k < HWORDS; k++) hist[k] = 0;
            for (int k = 2 * n; k >= 0; k--) { const int r = y + n - 2 - k; hist_push((r >= 0 && r < h) ? coln : 0u); }
            set_kf(y);
        }
        unsigned nib_c = 0;
        bool shortrow = false;
        if (!copyrow) {
            // ---- the previous output row joins the IIR sums ----
            if (y >= 1 && n >= 1) {
                Ent e[P];
                iir_entries(prev, e);
#pragma unroll
                for (int i = 0; i < P; i++) eadd(iirE[i], e[i]);
            }
            // ---- mask: the entering row's nibble goes in, the current and the leaving row's come out ----
            hist_push(nib_e);
            nib_c = hist_at(n - 1);
            const unsigned nib_l = hist_at(2 * n);
            if (rare_e) rare_rows++;
            if (y - n - 1 >= 0 && rare_any(nib_l)) rare_rows--;
            // sparse layer, no selected pixel in rows [y-n, y+n): the short path (needs neither FIR sums nor the image row)
            shortrow = SUMROW && sparse && rare_rows == 0;
            // ---- vertical running sums for row y (wave-uniform row tests) ----
            if (y + n - 1 < h && n >= 1) fir_apply(e_px, nib_e, true);        // ye = min(h, y+n)
            if (y - n - 1 >= 0 && n >= 1) {
                fir_apply(lv_px, nib_l, false);                                // ys = max(0, y-n)
                unsigned o[ND];
#pragma unroll
                for (int q = 0; q < ND; q++) o[q] = ol[q] & pxm[q];
                Ent e[P];
                iir_entries(o, e);
#pragma unroll
                for (int i = 0; i < P; i++) esub(iirE[i], e[i]);
            }
            if (y <= n) {
                asm volatile("" ::: "memory");        // a real (wave-uniform) branch: keeps the multiplies out of the steady state
                set_kf(y);
            }
        }
        // ---- the set's registers (but c) are free in every mode: the inputs of the row after the next, ONE site ----
        asm volatile("" : "+v"(firE[0].d[0]), "+v"(iirE[0].d[0]) : : "memory");      // (after the last use of e / lv / ol)
        request_top(S, y + 2);

        unsigned qd[ND];
#pragma unroll
        for (int q = 0; q < ND; q++) qd[q] = 0;
        if (copyrow) {
            wait_c(S, steady);
#pragma unroll
            for (int q = 0; q < ND; q++) prev[q] = px_dword<C>(S.c, q) & pxm[q];
        } else if (shortrow) {
            if constexpr (SUMROW) {
                // ---- short path: T(x) = sum of the IIR column sums of [x-n, x), count = (y-ys)(x-xs) ----
#pragma unroll
                for (int i = 0; i < P; i++) {
                    const int e = eidx(i);
                    if constexpr (EW == 2) *reinterpret_cast<uint2 *>(iirA + e) = make_uint2(iirE[i].d[0], iirE[i].d[1]);
                    else iirA[e] = iirE[i].d[0];
                }
                asm volatile("" ::: "memory");
                Ent aL;
#pragma unroll
                for (int k = 0; k < EW; k++) aL.d[k] = 0;
#pragma unroll
                for (int j = -NCT; j < 0; j++) eadd(aL, lds_ld(iirA, j));
#pragma unroll
                for (int i = 0; i < P; i++) {
                    float fsum[C];
                    if constexpr (C == 3) {
                        fsum[0] = (float)(aL.d[0] & 0xffffu); fsum[1] = (float)(aL.d[0] >> 16); fsum[2] = (float)(aL.d[EW - 1] & 0xffffu);
                    } else {
                        fsum[0] = (float)(aL.d[0] & 0xffffu);
                    }
                    const float qoff = __builtin_fmaf(rcK[i], 0.5f, -0.5f);
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const int jb = i * C + c;
                        qd[jb >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(fsum[c], rcK[i], qoff), jb & 3, qd[jb >> 2]);
                    }
                    if (i + 1 < P) { eadd(aL, iirE[i]); esub(aL, lds_ld(iirA, i - NCT)); }
                }
#pragma unroll
                for (int q = 0; q < ND; q++) prev[q] = qd[q] & pxm[q];
            }
        } else {
        // ---- publish: the registers already hold the LDS entry format ----
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int e = eidx(i);
            Ent second = iirE[i];
            if constexpr (SUMROW) eadd(second, firE[i]);
            if constexpr (EW == 2) {
                *reinterpret_cast<uint2 *>(firA + e) = make_uint2(firE[i].d[0], firE[i].d[1]);
                *reinterpret_cast<uint2 *>(iirA + e) = make_uint2(second.d[0], second.d[1]);
            } else {
                firA[e] = firE[i].d[0];
                iirA[e] = second.d[0];
            }
        }
        asm volatile("" ::: "memory");            // one wave: its LDS operations execute in order, no barrier, no wait

        // Only pixels with mask==0 get a quotient; a wave-row in which every core pixel is selected is a copy
        const unsigned on_cur = nib_bytes(nib_c);
        if (__any(act && on_cur != colm)) {
        Ent aL, aR, aI;
#pragma unroll
        for (int k = 0; k < EW; k++) { aL.d[k] = 0; aR.d[k] = 0; aI.d[k] = 0; }
        if constexpr (SUMROW) {
#pragma unroll
            for (int j = -NCT; j < 0; j++) eadd(aL, lds_ld(iirA, j));          // (fir + iir)[x0 + j]
#pragma unroll
            for (int j = 0; j < NCT; j++) eadd(aL, j < P ? firE[j < P ? j : 0] : lds_ld(firA, j));
        } else if constexpr (NCT >= 0) {
#pragma unroll
            for (int j = -NCT; j < 0; j++) { eadd(aL, lds_ld(firA, j)); eadd(aI, lds_ld(iirA, j)); }
#pragma unroll
            for (int j = 0; j < NCT; j++) {
                const Ent e = j < P ? firE[j < P ? j : 0] : lds_ld(firA, j);
                if constexpr (NH == 2) eadd(aR, e); else eadd(aL, e);
            }
        } else {
            for (int j = -n; j < 0; j++) { eadd(aL, lds_ld(firA, j)); eadd(aI, lds_ld(iirA, j)); }
            for (int j = 0; j < n; j++) eadd(aR, lds_ld(firA, j));
        }
#pragma unroll
        for (int i = 0; i < P; i++) {
            float fsum[C], fcnt;
            if constexpr (SUMROW) {
                const Ent T = aL;
                if constexpr (C == 3) {
                    fsum[0] = (float)(T.d[0] & 0xffffu); fsum[1] = (float)(T.d[0] >> 16);
                    fsum[2] = (float)(T.d[1] & 0xffffu); fcnt = (float)(T.d[1] >> 16);
                } else {
                    fsum[0] = (float)(T.d[0] & 0xffffu); fcnt = (float)(T.d[0] >> 16);
                }
            } else if constexpr (C == 3 && NH == 2) {
                fsum[0] = (float)ws_add_dw0(ws_add_w0w0(aL.d[0], aR.d[0]), aI.d[0]);
                fsum[1] = (float)ws_add_dw1(ws_add_w1w1(aL.d[0], aR.d[0]), aI.d[0]);
                fsum[2] = (float)ws_add_dw0(ws_add_w0w0(aL.d[1], aR.d[1]), aI.d[1]);
                fcnt = (float)ws_add_w1w1(aL.d[1], aR.d[1]);
            } else if constexpr (C == 3) {
                fsum[0] = (float)((aL.d[0] & 0xffffu) + (aI.d[0] & 0xffffu));
                fsum[1] = (float)((aL.d[0] >> 16) + (aI.d[0] >> 16));
                fsum[2] = (float)((aL.d[1] & 0xffffu) + (aI.d[1] & 0xffffu));
                fcnt = (float)(aL.d[1] >> 16);
            } else {
                unsigned f0 = (aL.d[0] & 0xffffu) + aI.d[0], c0 = aL.d[0] >> 16;
                if constexpr (NH == 2) { f0 += aR.d[0] & 0xffffu; c0 += aR.d[0] >> 16; }
                fsum[0] = (float)f0; fcnt = (float)c0;
            }
            // floor(v / cnt) = round-to-nearest-even((v + 0.5) / cnt - 0.5) -- exhaustive self-test in k_optimise.hip
            const float rc = __builtin_amdgcn_rcpf(__builtin_fmaxf(fcnt + kf[i], 1.0f));
            const float qoff = __builtin_fmaf(rc, 0.5f, -0.5f);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int jb = i * C + c;
                qd[jb >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(fsum[c], rc, qoff), jb & 3, qd[jb >> 2]);
            }
            if (i + 1 < P) {      // slide to pixel x+1 (own columns' entries come from registers)
                if constexpr (SUMROW) {
                    eadd(aL, (i + NCT < P) ? firE[(i + NCT < P) ? i + NCT : 0] : lds_ld(firA, i + NCT));
                    eadd(aL, iirE[i]);
                    esub(aL, lds_ld(iirA, i - NCT));
                } else {
                    if constexpr (NH == 2) {
                        eadd(aL, firE[i]); esub(aL, lds_ld(firA, i - n));
                        eadd(aR, lds_ld(firA, i + n)); esub(aR, firE[i]);
                    } else {
                        eadd(aL, lds_ld(firA, i + n)); esub(aL, lds_ld(firA, i - n));
                    }
                    eadd(aI, iirE[i]); esub(aI, lds_ld(iirA, i - n));
                }
            }
        }
        }
        // masked pixels keep the image value (new_img = np.copy(img)), the others take the quotient
        {
            wait_c(S, steady);
            const unsigned on = on_cur;
            if constexpr (C == 3) {
                const unsigned e0 = __builtin_amdgcn_perm(0u, on, 0x01000000u), e1 = __builtin_amdgcn_perm(0u, on, 0x02020101u),
                               e2 = __builtin_amdgcn_perm(0u, on, 0x03030302u);
                prev[0] = ((px_dword<C>(S.c, 0) & e0) | (qd[0] & ~e0)) & pxm[0];
                prev[1] = ((px_dword<C>(S.c, 1) & e1) | (qd[1] & ~e1)) & pxm[1];
                prev[2] = ((px_dword<C>(S.c, 2) & e2) | (qd[2] & ~e2)) & pxm[2];
            } else {
                prev[0] = ((px_dword<C>(S.c, 0) & on) | (qd[0] & ~on)) & pxm[0];
            }
        }
        }
        // ---- the current image row of the row after the next, ONE site (a short row never looked at its own) ----
        asm volatile("" : "+v"(prev[0]), "+v"(prev[ND - 1]) : : "memory");
        request_c(S, y + 2);
    };
    // rows 0 and 1 wait for everything (their predecessors are not whole rows), the rest with the counts above
    do_row(0, SA, std::false_type{});
    if (h > 1) do_row(1, SB, std::false_type{});
    for (int y = 2; y < h; y += 2) {
        do_row(y, SA, std::true_type{});
        if (y + 1 < h) do_row(y + 1, SB, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last requests (rows past the image) land before the registers are reused
    if (h >= 1) store_row(h - 1, prev);
}

#ifndef MRCHIP_WS_WAVES
#define MRCHIP_WS_WAVES 3
#endif

// grid = jobs * smax single-wave workgroups: job = block / smax, strip = block % smax (left strips first)
template <int C, int GENERIC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MRCHIP_WS_WAVES, 8)))
void optimise_ws_kernel(const OptJob *jobs, WsLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int job = blockIdx.x / L.smax, strip = blockIdx.x - job * L.smax;
    const OptJob J = jobs[job];
    if (strip >= J.ws_S) return;
    unsigned *lds = reinterpret_cast<unsigned *>(smem);
    if constexpr (GENERIC) ws_rows<C, -1>(J, lds, L, strip);
    else {
        if (J.n == 3) ws_rows<C, 3>(J, lds, L, strip);              // fg, mrc.py:413/415
        else ws_rows<C, 10>(J, lds, L, strip);                      // bg, mrc.py:447/449
    }
}

// strips of a job: HL + HR halo lanes, the core lanes shared out evenly
void ws_geometry(int w, int n, int *S, int *clb) {
    const int HL = (n + 3) >> 2, HR = (n + 2) >> 2;
    const int CL = 64 - HL - HR;
    const int w4 = cdiv(w, 4);
    *S = std::max(1, cdiv(w4, CL));
    *clb = cdiv(w4, *S);
}

// Opt-in (MRCHIP_OPT_WS=1): measured on MI355X this schedule does not beat the workgroup schedule of k_optimise.hip yet
// (256 page-layers of 4000x3000 in 10.5 ms against 5.65 ms -- a lone wave needs ~1.3 us per row of
// ~400 instructions, and with two register sets only three waves fit a SIMD).  It stays buildable and tested
// (tests/test_gpu_kernels.py runs the optimise vectors through it) because its parts are the plan for the next step.
bool ws_supported(int w, int h, int n_max, int n_min) {
    static const bool on = getenv("MRCHIP_OPT_WS") && atoi(getenv("MRCHIP_OPT_WS")) != 0;
    // (n >= 2: the leaving output row of row y + 2 is requested while row y is being made; it is row y + 1 - n)
    return on && n_min >= 2 && n_max <= 11 && h < 65536 && w >= 1;
}

// h_jobs: host copy (its ws_* fields are filled here), d_jobs: where it is uploaded to.  All jobs share w, h, c; every
// job has mbits.  Returns 0 / error.
int launch_optimise_ws(mrchip_ctx *ctx, hipStream_t s, OptJob *h_jobs, OptJob *d_jobs, int njobs, int w, int h, int c,
                       OptMail *mail, double alg) {
    int smax = 1;
    bool generic = false;
    size_t granules = 32;                                   // (first 256 bytes unused) -- counted in 8-byte units
    for (int i = 0; i < njobs; i++) {
        OptJob &j = h_jobs[i];
        ws_geometry(w, j.n, &j.ws_S, &j.ws_clb);
        j.ws_mail = granules;
        granules += std::max((size_t)std::max(j.ws_S - 1, 0) * h, (size_t)1) * ((j.n + 3) >> 2) * c;
        smax = std::max(smax, j.ws_S);
        if (j.n != 3 && j.n != 10) generic = true;
    }
    const size_t need = granules * sizeof(u64) + 4096;     // (strips without a left neighbour still issue their hand-off loads)
    if (need > mail->bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        TRY(mail->buf.alloc(ctx, need + need / 8));
        mail->bytes = need + need / 8;
        mail->epoch = 0;
    }
    if (mail->epoch == 0 || mail->epoch >= 0xfffe) {       // fresh buffer, or the 16-bit epoch is about to repeat
        HIP_TRY(hipMemsetAsync(mail->buf.p, 0, mail->bytes, s));
        mail->epoch = 0;
    }
    mail->epoch++;
    HIP_TRY(hipMemcpyAsync(d_jobs, h_jobs, (size_t)njobs * sizeof(OptJob), hipMemcpyHostToDevice, s));
    WsLaunch L;
    L.mail = mail->buf.as<u64>();
    L.tagbase = mail->epoch << 16;
    L.err = mail->err;
    L.smax = smax;
    int n_max = 0;
    for (int i = 0; i < njobs; i++) n_max = std::max(n_max, h_jobs[i].n);
    const int nent = 64 * 4 + 2 * n_max;
    // two rows of column-sum entries + the left halo's ring of n + 1 granules
    const size_t lds = 2 * (size_t)(nent + nent / 4 + 1) * ((c == 3) ? 8 : 4) + (size_t)(n_max + 1) * ((n_max + 3) / 4) * c * 4;
    const char *nm = c == 3 ? "optimise_rgb" : "optimise_gray";
    const long long blocks = (long long)njobs * smax;
    if (blocks > 0x7fffffffLL) { set_error("optimise: %lld strips in one launch", blocks); return MRCHIP_E_UNSUPPORTED; }
#define WS_LAUNCH(CC, GG)                                                                                        \
    LAUNCH(ctx, s, nm, alg, hipLaunchKernelGGL((optimise_ws_kernel<CC, GG>), dim3((unsigned)blocks), dim3(64), lds, s, d_jobs, L))
    if (c == 3) { if (generic) WS_LAUNCH(3, 1); else WS_LAUNCH(3, 0); }
    else { if (generic) WS_LAUNCH(1, 1); else WS_LAUNCH(1, 0); }
#undef WS_LAUNCH
    return 0;
}

}  // namespace mrchip
