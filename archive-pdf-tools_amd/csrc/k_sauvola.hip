// Sauvola adaptive threshold (reference: cython/sauvola.pyx:29-222, closed form in
// SURVEY.md 8a row a1) for gfx950.
//
// Decomposition: one 64-lane wave per tile (workgroup = 1 wave, so the
// write->read hand-off through LDS needs no cross-wave barrier).  A tile is a
// strip of CW = 64*K input columns (K adjacent columns per lane, loaded as one
// aligned K-byte vector) by `rows` output rows.  Each lane keeps the vertical
// window sums of its K columns (sum and sum of squares, int32 like the
// reference's `integral` arrays, pyx:64-65) in registers and slides them down
// the strip: +entering row, -leaving row.  Per output row the wave builds the
// exclusive prefix of the column sums across the strip (K-element serial prefix
// per lane + one DPP wave scan of the lane totals), parks it in LDS, and every
// output pixel takes its horizontal window as a difference of two prefix
// values.  Everything is integer (mod 2^32, exact because a window sum of
// squares stays below 2^32 for windows up to 257x257) until the final
// comparison, which is evaluated in fp64 in the reference's operation order
// with contraction off.
//
// Traffic per pixel: 1 B written + 1 B read x (strip-halo overlap CW/(CW-ww)) x (tile warm-up
// (th+wh)/th) x 3 (entering, centre and leaving use of a row; the two re-reads are 25 rows old and
// come back from L2 / Infinity Cache).  Algorithmic bytes: 2*w*h.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "mrchip_internal.h"

namespace mrchip {

// Per window-row-count constants of the table-driven decision (one record per nrows = 1..wh, count = ww * nrows):
// exact truncating divisions by `count` as one 32x32->hi multiply + shift each (host-checked magic numbers).
struct SauvolaRow {
    unsigned ms, mq;          // floor(S / count) = mulhi(S, ms) >> ss  for 0 <= S <= 255 count;  likewise Q <= 65025 count
    int ss, sq;
    unsigned c255, c65025;    // 255 * count, 65025 * count: window sums of the inverted image (255 - p)
    int ok;                   // magic numbers exist for this count
    int pad_;
};

struct SauvolaParams {
    int ww, wh;       // window
    int l, r, o, u;   // l=(ww+1)/2 r=ww/2 o=(wh+1)/2 u=wh/2  (pyx:76-79)
    double k, km1, k2;
    int flags;
    int two;          // output columns per tile
    int th;           // output rows per tile
    // table-driven decision (see sauvola_kernel): per-row magic numbers; fast == 0: quotients in fp64 for every pixel
    int fast;
    const SauvolaRow *rows;   // [wh + 1], indexed by nrows
};


// inclusive wave scan (64 lanes) with DPP row shifts + row broadcasts (GFX9)
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
    unsigned x = v;
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return x;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <int K>
struct Vec;
template <>
struct Vec<4> { using T = unsigned int; };
template <>
struct Vec<8> { using T = uint2; };
template <>
struct Vec<16> { using T = uint4; };

template <int K>
__device__ __forceinline__ void load_px(const uint8_t *p, unsigned (&w)[K / 4]) {
    using V = typename Vec<K>::T;
    V v = *reinterpret_cast<const V *>(p);
    if constexpr (K == 4) { w[0] = v; }
    if constexpr (K == 8) { w[0] = v.x; w[1] = v.y; }
    if constexpr (K == 16) { w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
}

// The reference's decision (pyx:143-153) for one pixel; returns `form`.  The whole of it runs in
// fp64, divisions included.  floor(N / c) for integers N < 2^32, 1 <= c < 2^17:
// (N + 0.5) / c is at least 0.5/c away from every integer while the fma below (N exact, rc = 1/c to
// ~2^-51, see rcp_nr, one rounding) is off by less than (N/c) * 2^-50 < 0.5/c, so the floor is exact.  mean
// and Q/count are then integer-valued doubles -- the very values the reference converts from its
// truncated integer quotients (pyx:144-145) -- mean*mean and the variance are exact, and the rest
// is the reference's own operation sequence.  Cheaper than integer quotients + conversions:
// 3 conversions instead of 4 and no correction steps.
// 1/c for the quotients below: the hardware reciprocal is a ~2^-26 seed (on its own it flipped a near-tie
// pixel in a fuzz case); one Newton step brings it to ~2^-51, six orders of magnitude inside the
// 0.5/N margin the floor needs, at a fifth of the cost of a correctly rounded division.
__device__ __forceinline__ double rcp_nr(double c) {
    const double r0 = __builtin_amdgcn_rcp(c);
    return __builtin_fma(r0, __builtin_fma(-c, r0, 1.0), r0);
}

__device__ __forceinline__ bool sauvola_form_dd(double Sd, double Qd, double pxd, double rc, double hrc, bool kpos,
                                                double km1, double k2) {
    const double mean = __builtin_floor(__builtin_fma(Sd, rc, hrc));
    const double qd = __builtin_floor(__builtin_fma(Qd, rc, hrc));
    const double mm = __dmul_rn(mean, mean);
    const double variance = __dadd_rn(qd, -mm);
    const double tmp = __dadd_rn(pxd, __dmul_rn(mean, km1));
    const double lhs = __dmul_rn(tmp, tmp);
    const double rhs = __dmul_rn(__dmul_rn(mm, k2), variance);
    const bool neg = tmp <= 0;
    return kpos ? (neg || (lhs <= rhs)) : (neg && (lhs >= rhs));
}

// ---- asynchronous row loads ----------------------------------------------------------------------------------
// The three input rows of an output row (entering, leaving, centre) are loaded PF rows ahead into fixed register
// slots.  hipcc cannot keep such loads in flight across the loop: with a variable number of stores between a load
// and its use it falls back to `s_waitcnt vmcnt(0)`, which also waits for the loads just issued -- a full memory
// latency in every row (measured: the kernel's time did not change when its instruction count was halved).  So the
// loads and their waits are written by hand: the load is an asm statement whose output the compiler believes
// ready, and every use is preceded by an asm `s_waitcnt vmcnt(N)` that takes the slot as an in/out operand (which
// also pins the order).  vmcnt retires in order, so N = the number of LOADS issued after the one needed; the
// compiler's own stores in between can only make the wait stricter, never too weak.  The listing is checked for
// spills (none) -- a spill or copy of a slot between its load and its wait would read stale registers.
template <int KD> struct Slot;
template <> struct Slot<1> { typedef unsigned T; };
template <> struct Slot<2> { typedef unsigned T __attribute__((ext_vector_type(2))); };
template <> struct Slot<4> { typedef unsigned T __attribute__((ext_vector_type(4))); };

template <int KD>
__device__ __forceinline__ void row_load_async(typename Slot<KD>::T &r, unsigned off, const uint8_t *base) {
    if constexpr (KD == 1) asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    if constexpr (KD == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
    if constexpr (KD == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
}
template <int N, class T>
__device__ __forceinline__ void rows_wait(T &a, T &b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N, class T>
__device__ __forceinline__ void rows_wait(T &a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int KD>
__device__ __forceinline__ unsigned slot_dword(const typename Slot<KD>::T &r, int q) {
    if constexpr (KD == 1) return r; else return r[q];
}

// Per wave: a strip of CW = 64*K input columns by `rows` output rows (tall tiles amortise the
// (wh-1)-row warm-up).  (An LDS ring of the last wh rows was tried: it removes the re-reads but
// caps the CU at ~10 waves and ran 2x slower -- occupancy is what hides this kernel's per-row chain.)
// BOTH: every job also thresholds the image 255 - p (the hOCR-box launch; a page launch has BOTH = false and does
// not carry the second polarity's registers).
template <int K, bool MULTI, bool FAST, bool BOTH, int PL = 32>
__global__ __launch_bounds__(64, (K == 8 ? 4 : 1)) void sauvola_kernel(SauvolaJob job1, const SauvolaJob *jobs,
                                                     SauvolaParams P) {
    constexpr int KD = K / 4;
#ifndef SAUVOLA_PF8
#define SAUVOLA_PF8 2
#endif
    constexpr int PF = (K == 8) ? SAUVOLA_PF8 : 2;     // rows in flight = unroll factor of the row loop
    // Prefix rows in LDS, transposed: strip column ci = K*t + i lives at [i][t + PL].  A wave's
    // accesses for one pixel index i are then consecutive dwords (conflict-free); the natural
    // [ci] order would put lanes 16 B apart = a 4-way bank conflict on every read.  PL lanes of
    // slack on both sides: halo lanes evaluate the formula on out-of-strip indices instead of
    // branching around it (their results are never stored).  K * PL columns of slack must cover half a window:
    // PL = 8 for the windows the pipeline uses (LDS per wave 5 KiB of prefix rows + 4 KiB of tables: 16 waves per CU
    // fit), 32 for the widest ones.
    constexpr int LS = 64 + 2 * PL;
    // (prefix of sums, prefix of sums of squares) side by side: one 8-byte LDS access per column end instead of two
    __shared__ uint2 EBuf[K * LS];
    auto pidx = [&](int ci) { const int c2 = ci + K * PL; return (c2 % K) * LS + c2 / K; };
    // FAST: the two products of the reference's decision that depend on the (integer) mean only, for every mean
    // 0..255, rounded exactly as the reference rounds them (pyx:147-150): mean * (k - 1) and (mean * mean) * k2.
    // The decision then is: integer mean (index) -> two LDS reads -> add, two multiplies, two compares.
    // (Measured: the tables save 3 VALU instructions per pixel but their per-lane LDS reads conflict on noisy images --
    // 20 % of the LDS cycles -- and the kernel comes out 3 % slower on the c3gray batch, 7 % faster on blurred pages;
    // computing the two products from the integer mean is the robust choice.  TABLES keeps the variant buildable.)
    constexpr bool TABLES = false;
    __shared__ double Tkm1[(FAST && TABLES) ? 256 : 1], Tk2[(FAST && TABLES) ? 256 : 1];
    if constexpr (FAST && TABLES) {
        for (int m = threadIdx.x; m < 256; m += 64) {
            const double md = (double)m;
            Tkm1[m] = __dmul_rn(md, P.km1);
            Tk2[m] = __dmul_rn(__dmul_rn(md, md), P.k2);
        }
        lds_wave_sync();
    }

    SauvolaJob job = MULTI ? jobs[blockIdx.z] : job1;
    const int w = job.w, h = job.h;
    const int X0 = blockIdx.x * P.two;
    const int Y0 = blockIdx.y * P.th;
    if (X0 >= w || Y0 >= h) return;
    const int lane = threadIdx.x;
    const int nout = min(P.two, w - X0);
    const int rows = min(P.th, h - Y0);
    const int l = P.l, r = P.r, o = P.o, u = P.u;

    // first input column, aligned down so that every lane's K-byte load is aligned
    int Xa = X0 - l + 1;
    {
        uintptr_t a = reinterpret_cast<uintptr_t>(job.src) + (intptr_t)Xa;
        Xa -= (int)(a & (uintptr_t)(K - 1));
    }
    const int c0 = Xa + K * lane;          // this lane's first column
    // per-byte validity mask of the K columns (columns outside [0,w) contribute 0)
    unsigned vmask[KD];
#pragma unroll
    for (int q = 0; q < KD; q++) {
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int c = c0 + 4 * q + b;
            if (c >= 0 && c < w) m |= 0xffu << (8 * b);
        }
        vmask[q] = m;
    }
    // all outputs of this strip see the full window width -> count is wave-uniform per row
    const bool full_cols = (X0 - l + 1 >= 0) && (X0 + nout - 1 + r <= w - 1);

    unsigned cs[K], cq[K];
#pragma unroll
    for (int i = 0; i < K; i++) { cs[i] = 0; cq[i] = 0; }

    auto acc_row = [&](const unsigned (&wv)[KD], bool plus) {
#pragma unroll
        for (int q = 0; q < KD; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                unsigned p = (wv[q] >> (8 * b)) & 0xffu;
                if (plus) { cs[4 * q + b] += p; cq[4 * q + b] += p * p; }
                else      { cs[4 * q + b] -= p; cq[4 * q + b] -= p * p; }
            }
        }
    };
    // raw row from global memory (global, not flat, address space): the row index is clamped into
    // the image and nothing touches the loaded registers here, so the load stays in flight until the
    // row is USED several iterations later (masking at load time would put an s_waitcnt right here).
    // Rows outside the image are skipped where they are used (wave-uniform tests).
    typedef const unsigned __attribute__((address_space(1))) *gc_u32p;
    // wave-uniform row base (scalar registers) + the lane's 32-bit byte offset: the load takes the
    // SGPR-base addressing mode and costs no 64-bit vector address arithmetic
    const uint8_t *srcA = job.src + Xa;
    const unsigned loff = (unsigned)(K * lane);
    auto gload = [&](int yy, unsigned (&wv)[KD]) {
        const int yc = min(max(yy, 0), h - 1);
        const uint8_t *rowp = srcA + (size_t)yc * job.src_pitch;          // uniform
        gc_u32p p = (gc_u32p)(rowp + loff);
#pragma unroll
        for (int q = 0; q < KD; q++) wv[q] = p[q];
    };
    auto acc_row_m = [&](const unsigned (&wv)[KD], bool plus) {
        unsigned m[KD];
#pragma unroll
        for (int q = 0; q < KD; q++) m[q] = wv[q] & vmask[q];
        acc_row(m, plus);
    };
    // warm-up: rows [Y0-o, Y0+u-1] (clipped) enter the sums, four loads in flight at a time
    {
        int yy = max(0, Y0 - o);
        const int ye = min(h, Y0 + u);
        for (; yy + 4 <= ye; yy += 4) {
            unsigned wv[4][KD];
#pragma unroll
            for (int t = 0; t < 4; t++) gload(yy + t, wv[t]);
#pragma unroll
            for (int t = 0; t < 4; t++) acc_row_m(wv[t], true);
        }
        for (; yy < ye; yy++) {
            unsigned wv[KD];
            gload(yy, wv);
            acc_row_m(wv, true);
        }
    }
    // three register queues, PF slots each: entering rows y+u, leaving rows y-o, centre rows y.
    // Every address is known in advance, so the loads run PF rows ahead of their use and the
    // serial chain of a row never waits for memory (leaving / centre rows come back from L2/MALL).
    typedef typename Slot<KD>::T slot_t;
    slot_t qe[PF], ql[PF], qc[PF];
    const bool invert = (P.flags & SAUVOLA_INVERT) != 0;
    auto aload = [&](int yy, slot_t &r) {
        const int yc = min(max(yy, 0), h - 1);
        row_load_async<KD>(r, loff, srcA + (size_t)yc * job.src_pitch);     // uniform row base + lane offset
    };
    // every load of the loop is issued in this order: e, l (after the column update), c (after the decision)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the warm-up's loads are the compiler's: start from zero
#pragma unroll
    for (int d = 0; d < PF; d++) {
        aload(Y0 + u + d, qe[d]);
        aload(Y0 - o + d, ql[d]);
        aload(Y0 + d, qc[d]);
    }
    constexpr int WAIT_EL = 3 * (PF - 1) + 1;     // loads issued after row y's e/l pair: c(y), then e, l, c of PF-1 rows
    constexpr int WAIT_C = 3 * (PF - 1) + 2;      // ... after c(y): the same PF-1 rows, plus e(y+PF), l(y+PF)

    unsigned ones_a = 0, ones_b = 0;
    const bool kpos = P.k >= 0;

    // The row loop is unrolled PF times so that each slot of the three queues is a fixed set of registers: row y
    // reads slot y mod PF and, once it is done with it, loads row y + PF into the same registers.  No register
    // moves, and a load has PF - 1 whole rows to land (a shifting queue makes every row wait for the load issued
    // one row earlier, whatever its depth: the shift itself needs the data).
    int cur_nrows = -1;
    SauvolaRow RW = {};
    auto do_row = [&](const int y, slot_t &qev, slot_t &qlv, slot_t &qcv) {
        rows_wait<WAIT_EL>(qev, qlv);
        unsigned ev[KD], lv[KD], cv[KD];
#pragma unroll
        for (int q = 0; q < KD; q++) { ev[q] = slot_dword<KD>(qev, q); lv[q] = slot_dword<KD>(qlv, q); }
        // entering row y+u and leaving row y-o together: with d = pe - pl and t = pe + pl per column,
        // S += d and Q += pe^2 - pl^2 = d * t (one signed 24-bit multiply-add); a row outside the image
        // contributes zeros (wave-uniform selects)
        {
            const bool has_e = y + u < h, has_l = y - o >= 0;
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const unsigned em = has_e ? (ev[q] & vmask[q]) : 0u, lm = has_l ? (lv[q] & vmask[q]) : 0u;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int pe = (int)((em >> (8 * b)) & 0xffu), pl = (int)((lm >> (8 * b)) & 0xffu);
                    const int d = pe - pl, t = pe + pl;
                    cs[4 * q + b] += (unsigned)d;
                    cq[4 * q + b] += (unsigned)__mul24(d, t);
                }
            }
        }
        // this slot's next rows (ev / lv are dead from here on).  The loads must be issued AFTER the wait for this
        // row's data: vmcnt counts in order, so a load issued before that wait is waited for as well (a full memory
        // latency in every row).  The barrier takes a result of the column update as an operand -- a bare "memory"
        // clobber orders memory operations only and lets the scheduler sink the (register-only) update below it.
        asm volatile("" : "+v"(cs[K - 1]), "+v"(cq[0]) : : "memory");
        aload(y + u + PF, qev);
        aload(y - o + PF, qlv);
        const int nrows = __builtin_amdgcn_readfirstlane(min(y + u, h - 1) - max(y - o, -1));
        if (nrows != cur_nrows) {          // per-count constants: scalar loads, only when the row count changes
            // constant address space: a uniform load from it is a scalar load (a plain global load would become a
            // vector load behind the kernel's own stores and put a full-latency vmcnt(0) into the row)
            typedef const SauvolaRow __attribute__((address_space(4))) *crow_p;
            const crow_p rp = (crow_p)(uintptr_t)P.rows + nrows;
            RW.ms = rp->ms; RW.mq = rp->mq; RW.ss = rp->ss; RW.sq = rp->sq;
            RW.c255 = rp->c255; RW.c65025 = rp->c65025; RW.ok = rp->ok;
            cur_nrows = nrows;
        }

        // exclusive prefix over the strip's columns
        unsigned ps[K], pqx[K];
        unsigned ts = 0, tq = 0;
#pragma unroll
        for (int i = 0; i < K; i++) { ps[i] = ts; pqx[i] = tq; ts += cs[i]; tq += cq[i]; }
        unsigned bs = wave_scan_incl(ts) - ts;
        unsigned bq = wave_scan_incl(tq) - tq;
        lds_wave_sync();                   // previous row's LDS reads are done
#pragma unroll
        for (int i = 0; i < K; i++) {
            EBuf[i * LS + lane + PL] = make_uint2(bs + ps[i], bq + pqx[i]);
        }
        lds_wave_sync();

        // wave-uniform count / reciprocal when every output of the strip has the full window width
        const unsigned ucount = (unsigned)(P.ww * nrows);
        rows_wait<WAIT_C>(qcv);            // the centre row, loaded PF rows ago
#pragma unroll
        for (int q = 0; q < KD; q++) cv[q] = slot_dword<KD>(qcv, q);

        unsigned outa[KD], outb[KD];
#pragma unroll
        for (int q = 0; q < KD; q++) { outa[q] = 0; outb[q] = 0; }

        // window sums and count of column i of this lane (S, Q exact integers mod 2^32)
        auto window = [&](int i, unsigned &S, unsigned &Q, unsigned &count) {
            const int c = c0 + i;
            const int ci = K * lane + i;
            const uint2 ea = EBuf[pidx(ci + r + 1)], eb = EBuf[pidx(ci - l + 1)];
            S = ea.x - eb.x; Q = ea.y - eb.y;
            count = ucount;
            if (!full_cols) {
                const int ncols = min(c + r, w - 1) - max(c - l + 1, 0) + 1;
                count = (unsigned)max(ncols * nrows, 1);
            }
        };
        // ---- general path: the reference's decision in fp64 in its own operation order (any k, R, window, count) ----
        // one reciprocal per row where the whole strip sees the full window width (the count is uniform then)
        auto general_row = [&]() {
            const double urcd = rcp_nr((double)ucount), uhrcd = 0.5 * urcd;
#pragma unroll
            for (int i = 0; i < K; i++) {
                unsigned S, Q, count;
                window(i, S, Q, count);
                double rcd = urcd, hrcd = uhrcd;
                if (!full_cols) { rcd = rcp_nr((double)count); hrcd = 0.5 * rcd; }
                const unsigned px = (cv[i / 4] >> (8 * (i & 3))) & 0xffu;
                const double Sd = (double)S, Qd = (double)Q, pxd = (double)px;
                const bool fa = sauvola_form_dd(Sd, Qd, pxd, rcd, hrcd, kpos, P.km1, P.k2);   // pyx:144-151
                outa[i / 4] |= fa ? (1u << (8 * (i & 3))) : 0u;
                if constexpr (BOTH) {
                    // the same window on the image 255-p (mrc.py:224, 235)
                    // sum(255-p) = 255 n - S, sum((255-p)^2) = 65025 n - 510 S + Q: integers below 2^32, exact in
                    // fp64 whatever the rounding of the fmas (three conversions saved)
                    const double cd = (double)count;
                    const double Sid = __builtin_fma(255.0, cd, -Sd);
                    const double Qid = __builtin_fma(-510.0, Sd, __builtin_fma(65025.0, cd, Qd));
                    const bool fb = sauvola_form_dd(Sid, Qid, 255.0 - pxd, rcd, hrcd, kpos, P.km1, P.k2);
                    outb[i / 4] |= fb ? (1u << (8 * (i & 3))) : 0u;
                }
            }
        };
        if constexpr (!FAST) {
            general_row();
        } else if (!(full_cols && RW.ok)) {
            general_row();           // strips at the left / right image border, or a count without magic numbers
        } else {
            // ---- table path: one count for the whole row ------------------------------------------------------------
            // mean = S / count and Q / count as exact integer quotients (magic-number multiplies, host-checked for
            // every dividend the window can produce); variance = Q/count - mean^2 as an integer (>= 0 by Cauchy-Schwarz
            // and floor monotonicity) converted once -- the very value the reference's fp64 subtraction of two integers
            // yields; mean * (k-1) and mean^2 * k2 from the tables.  What remains of pyx:147-151 is its own sequence:
            // tmp = px + mean*(k-1); lhs = tmp*tmp; rhs = (mean^2*k2) * variance; two compares.  Bit-exact by
            // construction: every fp64 operation is the reference's, on the reference's operands.
            // (k >= 0 here: the launcher sends negative k to the general kernel.)  The form bits of a lane's K pixels are
            // shifted into one register through the carry: v_addc(bits, bits, form) = 2 bits + form, one instruction per
            // pixel where select + or take two; pixels run K-1 .. 0 so that bit i is pixel i.
            auto form_tab = [&](unsigned S, unsigned Q, double pxd, unsigned &bits) {
                const unsigned mean = __builtin_amdgcn_ubfe(__umulhi(S, RW.ms), (unsigned)RW.ss, 8u);   // <= 255: a table index even in halo lanes
                const unsigned q = __umulhi(Q, RW.mq) >> RW.sq;
                const double vard = (double)(q - __umul24(mean, mean));
                double tkm1, tk2;
                if constexpr (TABLES) { tkm1 = Tkm1[mean]; tk2 = Tk2[mean]; }
                else {
                    const double meand = (double)mean;
                    tkm1 = __dmul_rn(meand, P.km1);
                    tk2 = __dmul_rn(__dmul_rn(meand, meand), P.k2);
                }
                const double tmp = __dadd_rn(pxd, tkm1);
                const double lhs = __dmul_rn(tmp, tmp);
                const double rhs = __dmul_rn(tk2, vard);
                // two ballots of plain compares OR-ed on the scalar side (the ballot of `a || b` goes through a VGPR bool)
                const unsigned long long form = __builtin_amdgcn_ballot_w64(tmp <= 0) | __builtin_amdgcn_ballot_w64(lhs <= rhs);
                asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(form) : "vcc");
            };
            unsigned bits_a = 0, bits_b = 0;
#pragma unroll
            for (int i = K - 1; i >= 0; i--) {
                unsigned S, Q, count;
                window(i, S, Q, count);
                const double pxd = (double)((cv[i / 4] >> (8 * (i & 3))) & 0xffu);
                form_tab(S, Q, pxd, bits_a);
                if constexpr (BOTH)       // the window on 255 - p: sums from S, Q and the count (integers below 2^32)
                    form_tab(RW.c255 - S, (Q + RW.c65025) - 510u * S, 255.0 - pxd, bits_b);
            }
#pragma unroll
            for (int q = 0; q < KD; q++) {          // bit i -> byte i (0/1)
                outa[q] = __umul24((bits_a >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u;
                if constexpr (BOTH) outb[q] = __umul24((bits_b >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u;
            }
        }
        // form -> stored value (pyx:153 `0 if formres else 1`, complemented for mrc.py:85's np.invert), columns
        // outside the strip's outputs cleared, set pixels counted
        bool any = false, all = true;
#pragma unroll
        for (int q = 0; q < KD; q++) {
            unsigned vm = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = c0 + 4 * q + b;
                const bool valid = (c >= X0) && (c < X0 + nout);
                any |= valid;
                all &= valid;
                vm |= valid ? (1u << (8 * b)) : 0u;
            }
            outa[q] = (invert ? outa[q] : ~outa[q]) & vm;
            ones_a += __builtin_popcount(outa[q]);
            if constexpr (BOTH) {
                outb[q] = (invert ? outb[q] : ~outb[q]) & vm;
                ones_b += __builtin_popcount(outb[q]);
            }
        }
        asm volatile("" : "+v"(outa[0]), "+v"(outa[KD - 1]) : : "memory");     // as above: after the last use of cv
        aload(y + PF, qcv);                // the slot's next centre row (cv is dead from here on)
        if (any) {
            // global (not flat) stores: a flat access also counts on lgkmcnt -- every LDS wait of the next row would
            // wait for it -- and may retire out of order with the global loads, which the counted vmcnt waits of the
            // row queues cannot tolerate (seen as rare wrong tiles before the address space was pinned)
            typedef uint8_t __attribute__((address_space(1))) *g_u8p;
            typedef unsigned __attribute__((address_space(1))) *g_u32p;
            g_u8p d = (g_u8p)(uintptr_t)(job.dst + (size_t)y * job.dst_pitch + c0);
            const bool aligned = (((uintptr_t)d) & 3u) == 0;
            if constexpr (K >= 8) {
                if (job.bits) {
                    // the same pixels at 1 bit each (the denoiser's input rows: `pack` is not run): the lane's K columns
                    // are K/8 whole bytes of the bit row -- strips start at multiples of 8 columns when bits are asked
                    // for, so a byte has one owner; columns outside the image are zero in outa
                    unsigned byte = 0;
#pragma unroll
                    for (int q = 0; q < KD; q++) byte |= (((outa[q] * 0x01020408u) >> 24) & 0xFu) << (4 * q);
                    g_u8p bp = (g_u8p)(uintptr_t)(job.bits + (size_t)y * job.bits_pitch + (c0 >> 3));
                    if constexpr (K == 8) bp[0] = (uint8_t)byte;
                    else *(unsigned short __attribute__((address_space(1))) *)bp = (unsigned short)byte;
                }
            }
            if (all && aligned) {
#pragma unroll
                for (int q = 0; q < KD; q++) ((g_u32p)d)[q] = outa[q];
            } else {
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const int c = c0 + i;
                    if (c >= X0 && c < X0 + nout) {
                        uint8_t v = (uint8_t)((outa[i / 4] >> (8 * (i & 3))) & 0xffu);
                        d[i] = v;
                    }
                }
            }
            if constexpr (BOTH) {
                g_u8p e = (g_u8p)(uintptr_t)(job.dst_inv + (size_t)y * job.dst_pitch + c0);
                if (all && aligned) {
#pragma unroll
                    for (int q = 0; q < KD; q++) ((g_u32p)e)[q] = outb[q];
                } else {
#pragma unroll
                    for (int i = 0; i < K; i++) {
                        const int c = c0 + i;
                        if (c >= X0 && c < X0 + nout)
                            e[i] = (uint8_t)((outb[i / 4] >> (8 * (i & 3))) & 0xffu);
                    }
                }
            }
        }
    };
    for (int yb = Y0; yb < Y0 + rows; yb += PF) {
#pragma unroll
        for (int d = 0; d < PF; d++)
            if (yb + d < Y0 + rows) do_row(yb + d, qe[d], ql[d], qc[d]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the queue's last loads (rows past the tile) land before the registers are reused
    if (job.counts) {
        unsigned a = wave_sum(ones_a);
        unsigned b = wave_sum(ones_b);
        if (lane == 0) {
            if (a) atomicAdd(&job.counts[0], a);
            if (b) atomicAdd(&job.counts[1], b);
        }
    }
}

// floor(n / c) == mulhi(n, m) >> sh for every 0 <= n <= nmax, with m = ceil(2^(32+sh) / c) < 2^32: true iff
// (m c - 2^(32+sh)) nmax < 2^(32+sh) (Granlund-Montgomery).  The largest shift whose multiplier fits 32 bits
// is the most accurate one, so only that one is tried.
static bool magic_for(unsigned long long c, unsigned long long nmax, unsigned *m, int *sh) {
    if (c < 2) return false;                              // 2^32 / 1 does not fit the multiplier
    for (int s = 31; s >= 0; s--) {
        const unsigned __int128 p = (unsigned __int128)1 << (32 + s);
        const unsigned __int128 mm = (p + c - 1) / c;
        if (mm >> 32) continue;
        const unsigned __int128 e = mm * c - p;
        if (e * nmax < p) { *m = (unsigned)mm; *sh = s; return true; }
        return false;
    }
    return false;
}

// Row tables are a function of the window only: built once per (ww, wh) and kept for the life of the process
// (a few hundred bytes each).  Returns the device copy, nullptr if it cannot be made.
static const SauvolaRow *row_table(mrchip_ctx *ctx, int ww, int wh, std::vector<SauvolaRow> *host_copy = nullptr) {
    struct Entry { int dev, ww, wh; SauvolaRow *d; std::vector<SauvolaRow> h; };
    static std::vector<Entry> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : cache)
        if (e.dev == ctx->device && e.ww == ww && e.wh == wh) { if (host_copy) *host_copy = e.h; return e.d; }
    Entry e;
    e.dev = ctx->device; e.ww = ww; e.wh = wh; e.d = nullptr;
    e.h.assign(wh + 1, SauvolaRow{});
    for (int nr = 1; nr <= wh; nr++) {
        SauvolaRow &r = e.h[nr];
        const unsigned long long c = (unsigned long long)ww * nr;
        r.c255 = (unsigned)(255ull * c); r.c65025 = (unsigned)(65025ull * c);
        r.ok = 65025ull * c <= 0xffffffffull && magic_for(c, 255ull * c, &r.ms, &r.ss) &&
               magic_for(c, 65025ull * c, &r.mq, &r.sq);
    }
    if (hipMalloc((void **)&e.d, e.h.size() * sizeof(SauvolaRow)) != hipSuccess) return nullptr;
    if (hipMemcpy(e.d, e.h.data(), e.h.size() * sizeof(SauvolaRow), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    cache.push_back(e);
    if (host_copy) *host_copy = e.h;
    return e.d;
}

template <int K>
static int launch_k(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *h_jobs, const SauvolaJob *d_jobs,
                    int njobs, SauvolaParams P, int maxw, int maxh, double alg_bytes) {
    constexpr int CW = 64 * K;
    P.two = (CW - P.ww - K) & ~3;
    bool want_bits = false;
    for (int i = 0; i < njobs; i++) want_bits = want_bits || h_jobs[i].bits != nullptr;
    if (want_bits) {
        if (K < 8) { set_error("sauvola: 1-bpp output needs the 8- or 16-column kernel (sauvola_writes_bits)"); return MRCHIP_E_ARG; }
        // strips start at whole stores of the bit rows: a byte per lane for K = 8, a 16-bit short per lane for K = 16 (with
        // 8-column alignment two strips of the K = 16 kernel would share a short and race on it)
        P.two &= ~(K - 1);
    }
    if (P.two < 4) {
        set_error("sauvola: window width %d too large for the %d-column strip", P.ww, CW);
        return MRCHIP_E_UNSUPPORTED;
    }
    // rows per tile: tall tiles amortise the (wh-1)-row warm-up; shrink only while the launch
    // would leave the chip short of waves
    int strips = cdiv(maxw, P.two);
    // measured flat between 32 and 512 rows per tile on 64-page batches; 1024 starves the chip
    int th = 256;
    while (th > 32 && (long long)strips * cdiv(maxh, th) * njobs < 8192) th >>= 1;
    if (const char *e = getenv("MRCHIP_SAUVOLA_TH")) { int v = atoi(e); if (v >= 8) th = v; }   // tuning knob
    P.th = th;
    dim3 grid(strips, cdiv(maxh, th), njobs);
    const char *nm = (njobs == 1 && !d_jobs) ? "sauvola" : (h_jobs[0].dst_inv ? "sauvola_boxes" : "sauvola");
    const bool single = njobs == 1 && !d_jobs;
    const bool both = h_jobs[0].dst_inv != nullptr;
    for (int i = 1; i < njobs; i++)
        if ((h_jobs[i].dst_inv != nullptr) != both) { set_error("sauvola: jobs with and without a second polarity in one launch"); return MRCHIP_E_ARG; }
    // slack lanes of the LDS prefix rows: 8 when K * 8 columns cover half the window (the pipeline's windows), else 32
    const bool small_pl = K <= 8 && P.l + K <= K * 8;
#define SAUVOLA_LAUNCH(MULTI_, FAST_, BOTH_)                                                                     \
    do {                                                                                                         \
        if (small_pl && K <= 8)                                                                                  \
            LAUNCH(ctx, s, nm, alg_bytes, hipLaunchKernelGGL((sauvola_kernel<K, MULTI_, FAST_, BOTH_, (K <= 8 ? 8 : 32)>), grid, \
                                                             dim3(64), 0, s, h_jobs[0], d_jobs, P));              \
        else                                                                                                     \
            LAUNCH(ctx, s, nm, alg_bytes, hipLaunchKernelGGL((sauvola_kernel<K, MULTI_, FAST_, BOTH_, 32>), grid, dim3(64), 0, s, \
                                                             h_jobs[0], d_jobs, P));                              \
    } while (0)
    // the integer-quotient decision pays on the page kernel (one polarity: 38 -> 30 VALU instructions per pixel, 128 blurred
    // pages 2.48 -> 2.33 ms, the noisy c3gray batch unchanged); the two-polarity box kernel measured 4 % slower with it
    // (same-box A/B), so the boxes keep the fp64 quotients unless MRCHIP_SAUVOLA_FAST=2 asks otherwise
    const bool fast = P.fast && (!both || P.fast >= 2);
    const int sel = (single ? 0 : 4) | (fast ? 2 : 0) | (both ? 1 : 0);
    switch (sel) {
        case 0: SAUVOLA_LAUNCH(false, false, false); break;
        case 1: SAUVOLA_LAUNCH(false, false, true); break;
        case 2: SAUVOLA_LAUNCH(false, true, false); break;
        case 3: SAUVOLA_LAUNCH(false, true, true); break;
        case 4: SAUVOLA_LAUNCH(true, false, false); break;
        case 5: SAUVOLA_LAUNCH(true, false, true); break;
        case 6: SAUVOLA_LAUNCH(true, true, false); break;
        default: SAUVOLA_LAUNCH(true, true, true); break;
    }
#undef SAUVOLA_LAUNCH
    return 0;
}

// Self-test of the fp64 quotient used above: floor(fma(N, rcp_nr(c), 0.5 rcp_nr(c))) against integer
// division for every divisor the kernel can see (1 .. 65792) and, per divisor, the dividends where a
// quotient is most fragile: k*c - 1, k*c, k*c + 1 for ~4096 values of k spread over [0, 2^32/c).
__global__ __launch_bounds__(256) void sauvola_div_selftest_kernel(unsigned long long *bad) {
    const unsigned c = blockIdx.x * 256 + threadIdx.x + 1;
    if (c > 65792u) return;
    const double rc = rcp_nr((double)c), hrc = 0.5 * rc;
    const unsigned long long kmax = 0xffffffffull / c;
    const unsigned long long step = kmax / 4096 + 1;
    unsigned long long nbad = 0;
    for (unsigned long long k = 0; k <= kmax; k += step) {
        for (int d = -1; d <= 1; d++) {
            const long long N = (long long)(k * c) + d;
            if (N < 0 || N > 0xffffffffll) continue;
            const double q = __builtin_floor(__builtin_fma((double)(unsigned)N, rc, hrc));
            if (q != (double)((unsigned long long)N / c)) nbad++;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int sauvola_div_selftest(mrchip_ctx *ctx, hipStream_t s, unsigned long long *d_bad) {
    HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
    hipLaunchKernelGGL(sauvola_div_selftest_kernel, dim3(cdiv(65792, 256)), dim3(256), 0, s, d_bad);
    HIP_TRY(hipGetLastError());
    return 0;
}

// columns per lane of the kernel a launch takes
static int sauvola_columns_per_lane(int maxw, int maxh, int ww) {
    static const int force_k = getenv("MRCHIP_SAUVOLA_K") ? atoi(getenv("MRCHIP_SAUVOLA_K")) : 0;   // tuning knob
    // 8 columns per lane halve the strip halo (452 of 512 columns are outputs instead of 200 of 256) at the
    // price of 128 VGPRs: measured 12 % faster on whole pages, 10 % slower on the short hOCR-box crops
    const bool page_like = maxw >= 1024 && maxh >= 256;
    if (ww <= 120 && force_k != 8 && !(page_like && force_k != 4)) return 4;
    return ww <= 360 ? 8 : 16;
}

// jobs: host array.  For njobs > 1 (or d_jobs != nullptr) the same array must
// already be resident at d_jobs.
int launch_sauvola_dev(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *jobs, const SauvolaJob *d_jobs,
                       int njobs, int ww, int wh, double k, double R, int flags) {
    if (njobs <= 0) return 0;
    if (njobs > MAX_GRID_Z) {        // one job per grid.z slice: big batches of small pages go in several launches
        for (int o = 0; o < njobs; o += MAX_GRID_Z)
            TRY(launch_sauvola_dev(ctx, s, jobs + o, d_jobs + o, std::min(MAX_GRID_Z, njobs - o), ww, wh, k, R, flags));
        return 0;
    }
    if (ww < 1 || wh < 1) { set_error("sauvola: window must be >= 1"); return MRCHIP_E_ARG; }
    if ((long long)ww * wh > 65792) {
        // S = sum of a window < 2^24 (24-bit multiplies, exact fp32) and Q < 2^32
        set_error("sauvola: window %dx%d exceeds the supported area (<= 65792 = 256x257)", ww, wh);
        return MRCHIP_E_UNSUPPORTED;
    }
    SauvolaParams P;
    P.ww = ww; P.wh = wh;
    P.l = (ww + 1) / 2; P.r = ww / 2; P.o = (wh + 1) / 2; P.u = wh / 2;
    P.k = k; P.km1 = k - 1; P.k2 = k * k / R / R;     // pyx:62
    P.flags = flags;
    // Table-driven decision (sauvola_kernel, FAST): exact integer quotients by the row's uniform count + LDS tables of
    // the mean-dependent fp64 products; every fp64 operation left is the reference's own, so the result is bit-exact by
    // construction for any k and R.  Rows whose count has no 32-bit magic numbers, and strips at the left / right image
    // border (per-lane counts), take the general fp64 path inside the same kernel.  MRCHIP_SAUVOLA_FAST=0 forces the
    // general path everywhere (the parity tests run both).
    // (Round 2 first tried a guarded fp32 comparison here: bit-exact with its fp64 arbiter, but no fewer instructions
    // than the fp64 path -- DESIGN.md 5; it is gone, git history has it.)
    const char *fast_env = getenv("MRCHIP_SAUVOLA_FAST");
    const int no_fast = fast_env && atoi(fast_env) == 0;
    P.rows = row_table(ctx, ww, wh);
    if (!P.rows) { set_error("sauvola: cannot allocate the row table"); return MRCHIP_E_NOMEM; }
    P.fast = (!no_fast && k >= 0) ? ((fast_env && atoi(fast_env) >= 2) ? 2 : 1) : 0;
    int maxw = 0, maxh = 0;
    double alg = 0;
    for (int i = 0; i < njobs; i++) {
        if (jobs[i].w > maxw) maxw = jobs[i].w;
        if (jobs[i].h > maxh) maxh = jobs[i].h;
        alg += (double)jobs[i].w * jobs[i].h * (jobs[i].dst_inv ? 4.0 : 2.0);
        if ((jobs[i].src_pitch & 15) || (jobs[i].dst_pitch & 3)) {
            set_error("sauvola: pitches must be multiples of 16 (src) / 4 (dst)");
            return MRCHIP_E_ARG;
        }
    }
    const int K = sauvola_columns_per_lane(maxw, maxh, ww);
    if (K == 4) return launch_k<4>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
    if (K == 8) return launch_k<8>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
    return launch_k<16>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
}

bool sauvola_writes_bits(int maxw, int maxh, int ww) { return sauvola_columns_per_lane(maxw, maxh, ww) >= 8; }

int launch_sauvola(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *jobs, int njobs,
                   int ww, int wh, double k, double R, int flags) {
    if (njobs == 1) return launch_sauvola_dev(ctx, s, jobs, nullptr, 1, ww, wh, k, R, flags);
    set_error("launch_sauvola: multi-job launches go through launch_sauvola_dev");
    return MRCHIP_E_ARG;
}

}  // namespace mrchip
