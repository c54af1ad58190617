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

// Per window-row-count constants (one record per nrows = 1..wh, count = ww * nrows): the exact truncating mean
// floor(S / count) as one 32x32->hi multiply + shift (host-checked magic number).
struct SauvolaRow {
    unsigned ms;              // floor(S / count) = mulhi(S, ms) >> ss  for 0 <= S <= 255 count
    int ss;
    unsigned c255, c65025;    // 255 * count, 65025 * count: window sums of the inverted image (255 - p)
    int ok;                   // a magic number exists for this count and 65026 * count < 2^32
    int pad_[3];
};

constexpr int SAUVOLA_PF = 2;        // rows in flight in the row queues of sauvola_tile

struct SauvolaParams {
    int ww, wh;       // window
    int l, r, o, u;   // l=(ww+1)/2 r=ww/2 o=(wh+1)/2 u=wh/2  (pyx:76-79)
    double k, km1, k2;
    int flags;
    int two;          // output columns per tile
    int th;           // output rows per tile
    int strips, ytiles;       // tiles of the largest job (table kernel: a workgroup's waves take consecutive tiles)
    const SauvolaRow *rows;   // [wh + 1], indexed by nrows
    // ---- table-driven decision (sauvola_tab_kernel) ----
    const unsigned short *tab;   // [256][tabW]: T2[mean][clamp(px - mean, dlo1, dhi1) - dlo1]
    int tab_bytes;               // multiple of 16
    int tabW, dlo1, dhi1;
    const uint4 *colrec;         // [ww + 1] {ms, ss, count, ok} for count = ncols * wh: strips at the left / right border
    int colrec_n, colrec_ok;
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
// a pointer the program knows to be wave-uniform, in scalar registers whatever the compiler's divergence analysis made of
// it (the "s" operands of the asm loads and stores)
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
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

// ---- the table-driven decision (k >= 0) ------------------------------------------------------------------------
// For an integer mean m = S / count (truncating, pyx:144) and a pixel px the reference's decision (pyx:143-151)
//     tmp = px + m (k - 1);  form = tmp <= 0  or  tmp tmp <= ((m m) k2) var,      var = Q / count - m m
// depends on the window only through the INTEGER var in 0 .. 65025, and fl(A var) is monotone in var for A >= 0:
// there is a smallest Vmin(m, px) with form <=> var >= Vmin (0 where tmp <= 0, "never" where no var <= vmax(m)
// satisfies it).  With T2 = Vmin + m m:   form  <=>  floor(Q / count) >= T2  <=>  Q >= count T2   -- no division
// of Q, no conversion, no fp64 operation.  The table is built on the device with the very predicate of the general path
// (sauvola_form_dd) by bisection over var, per (k, R), and checked exhaustively against it (every (m, px, var):
// mrchip_selftest_sauvola_table).  "always" is stored as 0, "never" as 65026 (count 65026 < 2^32 > any Q).
// Reachable variances: every pixel is <= 255, so Q <= 255 S, floor(Q / count) <= min(65025, 255 m + 254) and var <=
// vmax(m) = min(65025, 255 m + 254) - m m; entries whose Vmin lies above that are "never" (so T2 <= 65025 otherwise).  That bounds the band of (m, px) whose entry is
// neither constant to d = px - m in [dlo, dhi] (k = 0.34, R = 128: -86 .. 0), and the LDS copy is
//     tab[m][clamp(px - m, dlo - 1, dhi + 1) - (dlo - 1)]    (16 bits each; 45 KB for k = 0.34, 14 KB for k = 0.1)
// whose first / last column are the constants.  Per pixel: mul_hi + bfe (mean), sub, med3, mad + lshl_add (address),
// ds_read_u16, mul_u24, compare, addc.
__device__ __forceinline__ bool sauvola_pred(double mean, double px, double var, double km1, double k2) {
    // the general path's function on count = 1: S = mean, Q = var + mean^2 (exact), 1 / count = 1
    return sauvola_form_dd(mean, __dadd_rn(var, __dmul_rn(mean, mean)), px, 1.0, 0.5, true, km1, k2);
}
__device__ __forceinline__ int sauvola_vmax(int m) { return min(65025, 255 * m + 254) - m * m; }

// full[m][px] = T2 (0 always, 65026 never)
__global__ __launch_bounds__(256) void sauvola_t2_build_kernel(unsigned short *full, double km1, double k2) {
    const int m = blockIdx.x, px = threadIdx.x;
    const int vmax = sauvola_vmax(m);
    int lo = 0, hi = vmax + 1;                 // smallest var in [0, vmax] with pred, vmax + 1 if none
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sauvola_pred((double)m, (double)px, (double)mid, km1, k2)) hi = mid; else lo = mid + 1;
    }
    full[m * 256 + px] = (unsigned short)(lo == 0 ? 0 : (lo > vmax ? 65026 : lo + m * m));
}

// every (m, px, var <= vmax(m)): the compact table (read with the kernel's own index arithmetic) against the predicate
__global__ __launch_bounds__(256) void sauvola_t2_selftest_kernel(const unsigned short *tab, int W, int dlo1, int dhi1,
                                                                  double km1, double k2, unsigned long long *bad,
                                                                  unsigned long long *tested) {
    const int m = blockIdx.x, px = threadIdx.x;
    const int d = px - m;
    const int j = min(max(d, dlo1), dhi1) - dlo1;
    const unsigned T2 = tab[m * W + j];
    const int vmax = sauvola_vmax(m);
    unsigned long long nbad = 0;
    for (int var = 0; var <= vmax; var++) {
        const bool want = sauvola_pred((double)m, (double)px, (double)var, km1, k2);
        const bool got = (unsigned)(var + m * m) >= T2;
        nbad += want != got;
    }
    if (nbad) atomicAdd(bad, nbad);
    atomicAdd(tested, (unsigned long long)(vmax + 1));
}

// One tile: a strip of CW = 64*K input columns by `rows` output rows on one wave (tall tiles amortise the
// (wh-1)-row warm-up).  (An LDS ring of the last wh rows was tried: it removes the re-reads but
// caps the CU at ~10 waves and ran 2x slower -- occupancy is what hides this kernel's per-row chain.)
// BOTH: every job also thresholds the image 255 - p (the hOCR-box launch; a page launch has BOTH = false and does
// not carry the second polarity's registers).
// TAB: rows whose count has a magic number take the table-driven decision above -- with scalar operands where the
// whole strip sees the full window width, with per-column records from LDS on strips at the left / right border
// while the window has its full height; every other row (and the whole of a TAB = false launch: k < 0, a table
// too large for LDS, MRCHIP_SAUVOLA_FAST=0) takes the reference's fp64 sequence.
// ebase: LDS byte offset of this wave's prefix rows; tab_lds / rec_lds: of the workgroup's tables.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 __attribute__((address_space(3))) *lds_u2p;
typedef u32x4 __attribute__((address_space(3))) *lds_u4p;
typedef unsigned short __attribute__((address_space(3))) *lds_u16p;

template <int K, bool TAB, bool BOTH, int PL, int NST = 0>
__device__ __forceinline__ void sauvola_tile(const SauvolaJob &job, const SauvolaParams &P, const unsigned ebase,
                                             const unsigned tab_lds, const unsigned rec_lds, const int X0, const int Y0,
                                             const int lane) {
    constexpr int KD = K / 4;
    constexpr int PF = SAUVOLA_PF;                     // rows in flight = unroll factor of the row loop
    // Prefix rows in LDS, transposed: strip column ci = K*t + i lives at [i][t + PL].  A wave's
    // accesses for one pixel index i are then consecutive 8-byte units (conflict-free); the natural
    // [ci] order would put lanes 16 B apart = a 4-way bank conflict on every read.  PL lanes of
    // slack on both sides: halo lanes evaluate the formula on out-of-strip indices instead of
    // branching around it (their results are never stored).  K * PL columns of slack must cover half a window:
    // PL = 8 for the windows the pipeline uses, 32 for the widest ones.
    // (prefix of sums, prefix of sums of squares) side by side: one 8-byte LDS access per column end instead of two
    constexpr int LS = 64 + 2 * PL;
    const int w = job.w, h = job.h;
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

    // LDS addresses: this lane's slot of prefix row i is wbase + i * LS * 8; the two window ends of pixel i are
    // strip columns ci + r + 1 and ci - l + 1 = (row (i + a) % K, slot lane + PL + a / K): lane part + a uniform part
    const unsigned wbase = ebase + (unsigned)(lane + PL) * 8u;
    auto uoff = [&](int i, int a) {                    // uniform: byte offset of strip column K*lane + i + a relative to wbase
        const int c2 = i + a + K * PL;                 // >= 0: K * PL covers half a window
        return (unsigned)(((c2 % K) * LS + c2 / K - PL) * 8);
    };

    // window column sums while the tile warms up, then (in place) the exclusive prefix over the strip's columns
    unsigned cs[K], cq[K];
#pragma unroll
    for (int i = 0; i < K; i++) { cs[i] = 0; cq[i] = 0; }

    // raw row from global memory (global, not flat, address space): the row index is clamped into
    // the image; rows outside the image are skipped where they are used (wave-uniform tests).
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
    auto acc_row_m = [&](const unsigned (&wv)[KD]) {
#pragma unroll
        for (int q = 0; q < KD; q++) {
            const unsigned m = wv[q] & vmask[q];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned p = (m >> (8 * b)) & 0xffu;
                cs[4 * q + b] += p; cq[4 * q + b] += p * p;
            }
        }
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
            for (int t = 0; t < 4; t++) acc_row_m(wv[t]);
        }
        for (; yy < ye; yy++) {
            unsigned wv[KD];
            gload(yy, wv);
            acc_row_m(wv);
        }
    }
    // column sums -> exclusive prefix over the strip (lane-local prefix + wave scan of the lane totals).  From here on
    // the registers hold the PREFIX: a row adds the prefix of its column deltas (one three-operand add per value)
    // instead of updating K sums and prefixing them again.
    {
        unsigned ts = 0, tq = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            const unsigned a = cs[i], b = cq[i];
            cs[i] = ts; cq[i] = tq; ts += a; tq += b;
        }
        const unsigned bs = wave_scan_incl(ts) - ts, bq = wave_scan_incl(tq) - tq;
#pragma unroll
        for (int i = 0; i < K; i++) { cs[i] += bs; cq[i] += bq; }
    }
    // three register queues, PF slots each: entering rows y+u, leaving rows y-o, centre rows y.
    // Every address is known in advance, so the loads run PF rows ahead of their use and the
    // serial chain of a row never waits for memory (leaving / centre rows come back from L2/MALL).
    typedef typename Slot<KD>::T slot_t;
    slot_t qe[PF], ql[PF], qc[PF];
    const bool invert = (P.flags & SAUVOLA_INVERT) != 0;
    auto aload = [&](int yy, slot_t &rr) {
        const int yc = min(max(yy, 0), h - 1);
        row_load_async<KD>(rr, loff, srcA + (size_t)yc * job.src_pitch);     // uniform row base + lane offset
    };
    // every load of the loop is issued in this order: e, l (after the column update), c (after the decision)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the warm-up's loads are the compiler's: start from zero
    // counted stores: the lane's K output bytes as one vector store (lanes are wholly inside or wholly outside the
    // strip's outputs: strips start and end at multiples of K columns), one byte of the 1-bpp row; both written as asm
    // under an exec mask, so that a row issues exactly NST store instructions whatever its lanes do
    const bool lane_valid = (c0 >= X0) && (c0 < X0 + nout);
    const unsigned long long valid_mask = __builtin_amdgcn_ballot_w64(lane_valid);
    const uint8_t *dstA = NST > 0 ? job.dst + Xa : nullptr;
    const uint8_t *bitsA = NST > 1 ? job.bits + (Xa >> 3) : nullptr;        // Xa is a multiple of 8 (may be negative)
    auto store_counted = [&](const uint8_t *rowp, const uint8_t *brow, const unsigned (&ov)[KD], unsigned obits) {
        static_assert(NST == 0 || KD == 2, "counted stores: the 8-column kernel");
        if constexpr (NST > 0) {
            unsigned long long saved;
            rowp = uniform_ptr(rowp); brow = uniform_ptr(brow);
            if constexpr (NST != 3) {         // NST: 1 = the bytes, 2 = bytes and bit row, 3 = the bit row only (SauvolaJob::no_bytes)
                u32x2 dv = u32x2{ov[0], ov[KD - 1]};
                asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %3, %4\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(valid_mask), "v"(loff), "v"(dv), "s"(rowp) : "memory", "scc");
            }
            if constexpr (NST > 1) {
                const unsigned boff = (unsigned)lane;
                asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_byte %2, %3, %4\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(valid_mask), "v"(boff), "v"(obits), "s"(brow) : "memory", "scc");
            }
        }
    };
    // bit i: the lane's column i is one of the strip's outputs (and inside the image)
    unsigned colbits = 0;
#pragma unroll
    for (int i = 0; i < K; i++) colbits |= ((c0 + i >= X0) && (c0 + i < X0 + nout)) ? (1u << i) : 0u;
#pragma unroll
    for (int d = 0; d < PF; d++) {
        aload(Y0 + u + d, qe[d]);
        aload(Y0 - o + d, ql[d]);
        aload(Y0 + d, qc[d]);
        if constexpr (NST > 0) {
            // the steady state has the stores of an earlier row here: the same number of (harmless) stores to the tile's
            // first row, which row Y0 itself overwrites later (stores of one wave to one address stay in order)
            const unsigned zero[KD] = {};
            store_counted(dstA + (size_t)Y0 * job.dst_pitch, NST > 1 ? bitsA + (size_t)Y0 * job.bits_pitch : nullptr, zero, 0u);
        }
    }
    // NST > 0 (counted stores, see row_store_masked): every row issues exactly NST store instructions after its three
    // loads, so the stores count too -- PF rows of them lie between a load and its wait.  With NST = 0 the row's stores
    // are the compiler's (a varying number): leaving them out of N makes the wait stricter than needed -- it then also
    // waits for the acknowledgement of stores two rows old and for loads issued one row ago, which is what the
    // counted stores are for (128 pages 4000x3000: 1.93 -> 1.66 ms with the stores removed altogether).
    constexpr int NSTC = NST == 2 ? 2 : (NST ? 1 : 0);     // store INSTRUCTIONS per row
    constexpr int WAIT_EL = 3 * (PF - 1) + 1 + NSTC * PF;     // vm operations issued after row y's e/l pair: c(y), the stores of row y-PF, then PF-1 whole rows
    constexpr int WAIT_C = 3 * (PF - 1) + 2 + NSTC * PF;      // ... after c(y): the same, plus e(y+PF), l(y+PF)

    unsigned ones_a = 0, ones_b = 0;
    const bool kpos = P.k >= 0;

    // table path: clamp bounds and the address constant in registers (a VALU instruction reads one scalar operand)
    int dlo1v = P.dlo1, dhi1v = P.dhi1;
    unsigned tabC = tab_lds - 2u * (unsigned)P.dlo1;
    const unsigned tabW2 = 2u * (unsigned)P.tabW;
    if constexpr (TAB) asm volatile("" : "+v"(dlo1v), "+v"(dhi1v), "+v"(tabC));
    // strips at the left / right border: LDS address of the record of each of the lane's columns (count = ncols * wh)
    unsigned recaddr[K];
    if constexpr (TAB) {
#pragma unroll
        for (int i = 0; i < K; i++) {
            const int c = c0 + i;
            const int ncols = min(c + r, w - 1) - max(c - l + 1, 0) + 1;
            recaddr[i] = rec_lds + 16u * (unsigned)min(max(ncols, 1), P.ww);
        }
    }

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
        // entering row y+u and leaving row y-o together: with d = pe - pl and t = pe + pl per column the column's
        // S changes by d and its Q by pe^2 - pl^2 = d * t (one signed 24-bit multiply-add); es / eq = exclusive prefix
        // of those deltas over the lane's columns, rs / rq their totals.  A row outside the image contributes zeros
        // (wave-uniform selects)
        unsigned es[K], eq[K], rs = 0, rq = 0;
        {
            const bool has_e = y + u < h, has_l = y - o >= 0;
#pragma unroll
            for (int q = 0; q < KD; q++) {
                const unsigned em = has_e ? (ev[q] & vmask[q]) : 0u, lm = has_l ? (lv[q] & vmask[q]) : 0u;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int pe = (int)((em >> (8 * b)) & 0xffu), pl = (int)((lm >> (8 * b)) & 0xffu);
                    const int d = pe - pl, t = pe + pl;
                    es[4 * q + b] = rs; eq[4 * q + b] = rq;
                    rs += (unsigned)d;
                    rq += (unsigned)__mul24(d, t);
                }
            }
        }
        // this slot's next rows (ev / lv are dead from here on).  The loads must be issued AFTER the wait for this
        // row's data: vmcnt counts in order, so a load issued before that wait is waited for as well (a full memory
        // latency in every row).  The barrier takes a result of the column update as an operand -- a bare "memory"
        // clobber orders memory operations only and lets the scheduler sink the (register-only) update below it.
        asm volatile("" : "+v"(rs), "+v"(rq) : : "memory");
        aload(y + u + PF, qev);
        aload(y - o + PF, qlv);
        const int nrows = __builtin_amdgcn_readfirstlane(min(y + u, h - 1) - max(y - o, -1));
        if (nrows != cur_nrows) {          // per-count constants: scalar loads, only when the row count changes
            // constant address space: a uniform load from it is a scalar load (a plain global load would become a
            // vector load behind the kernel's own stores and put a full-latency vmcnt(0) into the row)
            typedef const SauvolaRow __attribute__((address_space(4))) *crow_p;
            const crow_p rp = (crow_p)(uintptr_t)P.rows + nrows;
            RW.ms = rp->ms; RW.ss = rp->ss; RW.c255 = rp->c255; RW.c65025 = rp->c65025; RW.ok = rp->ok;
            cur_nrows = nrows;
        }

        // the row's prefix = previous prefix + prefix of the deltas (lane-local part + wave scan of the lane totals)
        const unsigned dbs = wave_scan_incl(rs) - rs;
        const unsigned dbq = wave_scan_incl(rq) - rq;
        lds_wave_sync();                       // previous row's LDS reads are done
#pragma unroll
        for (int i = 0; i < K; i++) {
            cs[i] = cs[i] + es[i] + dbs;
            cq[i] = cq[i] + eq[i] + dbq;
            *(lds_u2p)(uintptr_t)(wbase + (unsigned)(i * LS * 8)) = u32x2{cs[i], cq[i]};
        }
        lds_wave_sync();

        // wave-uniform count when every output of the strip has the full window width
        const unsigned ucount = (unsigned)(P.ww * nrows);
        rows_wait<WAIT_C>(qcv);            // the centre row, loaded PF rows ago
#pragma unroll
        for (int q = 0; q < KD; q++) cv[q] = slot_dword<KD>(qcv, q);

        unsigned fbits_a = 0, fbits_b = 0;        // bit i: `form` of the lane's column i (second polarity: fbits_b)

        // window sums of column i of this lane (S, Q exact integers mod 2^32)
        auto window = [&](int i, unsigned &S, unsigned &Q) {
            const u32x2 ea = *(lds_u2p)(uintptr_t)(wbase + uoff(i, r + 1)), eb = *(lds_u2p)(uintptr_t)(wbase + uoff(i, 1 - l));
            S = ea.x - eb.x; Q = ea.y - eb.y;
        };
        // ---- general path: the reference's decision in fp64 in its own operation order (any k, R, window, count) ----
        // one reciprocal per row where the whole strip sees the full window width (the count is uniform then)
        auto general_row = [&]() {
            const double urcd = rcp_nr((double)ucount), uhrcd = 0.5 * urcd;
#pragma unroll
            for (int i = 0; i < K; i++) {
                unsigned S, Q, count = ucount;
                window(i, S, Q);
                double rcd = urcd, hrcd = uhrcd;
                if (!full_cols) {
                    const int c = c0 + i;
                    const int ncols = min(c + r, w - 1) - max(c - l + 1, 0) + 1;
                    count = (unsigned)max(ncols * nrows, 1);
                    rcd = rcp_nr((double)count); hrcd = 0.5 * rcd;
                }
                const unsigned px = (cv[i / 4] >> (8 * (i & 3))) & 0xffu;
                const double Sd = (double)S, Qd = (double)Q, pxd = (double)px;
                const bool fa = sauvola_form_dd(Sd, Qd, pxd, rcd, hrcd, kpos, P.km1, P.k2);   // pyx:144-151
                fbits_a |= fa ? (1u << i) : 0u;
                if constexpr (BOTH) {
                    // the same window on the image 255-p (mrc.py:224, 235)
                    // sum(255-p) = 255 n - S, sum((255-p)^2) = 65025 n - 510 S + Q: integers below 2^32, exact in
                    // fp64 whatever the rounding of the fmas (three conversions saved)
                    const double cd = (double)count;
                    const double Sid = __builtin_fma(255.0, cd, -Sd);
                    const double Qid = __builtin_fma(-510.0, Sd, __builtin_fma(65025.0, cd, Qd));
                    const bool fb = sauvola_form_dd(Sid, Qid, 255.0 - pxd, rcd, hrcd, kpos, P.km1, P.k2);
                    fbits_b |= fb ? (1u << i) : 0u;
                }
            }
        };
        // ---- table path ----
        // form <=> Q >= count * T2[mean][px]; the form bits of a lane's K pixels are shifted into one register through
        // the carry: v_addc(bits, bits, form) = 2 bits + form; pixels run K-1 .. 0 so that bit i is pixel i.
        // Three phases so that the LDS round trips overlap: every window end first, then every table entry, then the
        // compares (a pixel at a time the row is a chain of 2 x 8 dependent LDS latencies).
        auto t2_entry = [&](unsigned S, unsigned px, unsigned ms, unsigned ss) {
            const unsigned mean = __builtin_amdgcn_ubfe(__umulhi(S, ms), ss, 8u);   // <= 255: a table row even in halo lanes
            const int d = (int)px - (int)mean;
            int j;
            asm("v_med3_i32 %0, %1, %2, %3" : "=v"(j) : "v"(d), "v"(dlo1v), "v"(dhi1v));
            unsigned rowaddr = __umul24(mean, tabW2) + tabC;         // v_mad_u32_u24
            asm("" : "+v"(rowaddr));                                 // (keeps the compiler from re-associating it into mul + add3)
            const unsigned addr = ((unsigned)j << 1) + rowaddr;      // v_lshl_add_u32
            return (unsigned)*(lds_u16p)(uintptr_t)addr;
        };
        auto table_row = [&](auto per_lane) {
            constexpr bool PER_LANE = decltype(per_lane)::value;
            unsigned Sv[K], Qv[K], Ta[K], Tb[K], cntv[K];
#pragma unroll
            for (int i = 0; i < K; i++) window(i, Sv[i], Qv[i]);
#pragma unroll
            for (int i = 0; i < K; i++) {
                const unsigned px = (cv[i / 4] >> (8 * (i & 3))) & 0xffu;
                unsigned ms = RW.ms, ss = (unsigned)RW.ss, c255 = RW.c255;
                cntv[i] = ucount;
                if constexpr (PER_LANE) {
                    const u32x4 rec = *(lds_u4p)(uintptr_t)recaddr[i];
                    ms = rec.x; ss = rec.y; cntv[i] = rec.z;
                    if constexpr (BOTH) c255 = __umul24(cntv[i], 255u);
                }
                Ta[i] = t2_entry(Sv[i], px, ms, ss);
                if constexpr (BOTH) Tb[i] = t2_entry(c255 - Sv[i], 255u - px, ms, ss);   // the window on 255 - p (mrc.py:224, 235)
            }
            unsigned bits_a = 0, bits_b = 0;
#pragma unroll
            for (int i = K - 1; i >= 0; i--) {
                const unsigned long long fa = __builtin_amdgcn_ballot_w64(Qv[i] >= __umul24(cntv[i], Ta[i]));
                asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(bits_a) : "s"(fa) : "vcc");
                if constexpr (BOTH) {       // sum((255-p)^2) = 65025 n - 510 S + Q (an integer below 2^32)
                    const unsigned c65025 = PER_LANE ? __umul24(cntv[i], 65025u) : RW.c65025;
                    const unsigned Q2 = (Qv[i] + c65025) - 510u * Sv[i];
                    const unsigned long long fb = __builtin_amdgcn_ballot_w64(Q2 >= __umul24(cntv[i], Tb[i]));
                    asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(bits_b) : "s"(fb) : "vcc");
                }
            }
            fbits_a = bits_a; fbits_b = bits_b;
        };
        if constexpr (!TAB) {
            general_row();
        } else if (full_cols && RW.ok) {
            table_row(std::false_type{});
        } else if (!full_cols && nrows == P.wh && P.colrec_ok) {
            table_row(std::true_type{});
        } else {
            general_row();           // a count without a magic number; border strips while the window is clipped vertically
        }
        // form -> stored value (pyx:153 `0 if formres else 1`, complemented for mrc.py:85's np.invert), columns
        // outside the strip's outputs cleared, set pixels counted: all on the K form bits, then bit i -> byte i (0 / 1)
        const unsigned obits_a = (invert ? fbits_a : ~fbits_a) & colbits;
        ones_a += __builtin_popcount(obits_a);
        unsigned outa[KD], outb[KD];
#pragma unroll
        for (int q = 0; q < KD; q++) outa[q] = __umul24((obits_a >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u;
        unsigned obits_b = 0;
        if constexpr (BOTH) {
            obits_b = (invert ? fbits_b : ~fbits_b) & colbits;
            ones_b += __builtin_popcount(obits_b);
#pragma unroll
            for (int q = 0; q < KD; q++) outb[q] = __umul24((obits_b >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u;
        }
        const bool any = colbits != 0, all = colbits == ((1u << K) - 1u);
        asm volatile("" : "+v"(outa[0]), "+v"(outa[KD - 1]) : : "memory");     // as above: after the last use of cv
        aload(y + PF, qcv);                // the slot's next centre row (cv is dead from here on)
        if constexpr (NST > 0) {
            store_counted(dstA + (size_t)y * job.dst_pitch, NST > 1 ? bitsA + (size_t)y * job.bits_pitch : nullptr, outa, obits_a);
        } else if (any) {
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
                    const unsigned byte = obits_a;
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
    // A row past the tile's last one LEAVES the loop (it is not skipped inside it): every path through the listing is
    // then one the program can take.  tools/isa_vmflow.py follows each asm load to its wait along every path of the
    // control-flow graph; `if (yb + d < end) do_row(..)` inside a `for (yb < end)` loop gave it the path "slot 0, skip
    // slot 1, slot 0 again", which no tile runs.
    {
        static_assert(PF == 2, "the row loop is written out for two slots");
        const int yend = Y0 + rows;
        for (int yb = Y0; yb < yend; yb += PF) {
            do_row(yb, qe[0], ql[0], qc[0]);
            if (yb + 1 >= yend) break;
            do_row(yb + 1, qe[1], ql[1], qc[1]);
        }
    }
    // The queue's last loads (rows past the tile) land before their registers are reused: the wait takes every slot as
    // an in/out operand, so the compiler keeps the slots allocated up to it -- a bare asm with a "memory" clobber lets the
    // register allocator hand a dead slot to an ordinary value BEFORE the wait, which the landing load then overwrites.
    static_assert(PF == 2, "the closing wait names the slots of two rows");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qe[0]), "+v"(ql[0]), "+v"(qc[0]), "+v"(qe[1]), "+v"(ql[1]), "+v"(qc[1]) : : "memory");
    if (job.counts) {
        unsigned a = wave_sum(ones_a);
        unsigned b = wave_sum(ones_b);
        if (lane == 0) {
            if (a) atomicAdd(&job.counts[0], a);
            if (b) atomicAdd(&job.counts[1], b);
        }
    }
}

// LDS byte offset of a __shared__ object (the low half of its flat address)
template <class T>
__device__ __forceinline__ unsigned lds_offset(T *p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)p;
}

// general kernel: one wave per workgroup, one tile per wave, the fp64 decision everywhere
template <int K, bool MULTI, bool BOTH, int PL = 32>
__global__ __launch_bounds__(64, (K == 8 ? 4 : 1)) void sauvola_kernel(SauvolaJob job1, const SauvolaJob *jobs,
                                                     SauvolaParams P) {
    constexpr int LS = 64 + 2 * PL;
    __shared__ uint2 EBuf[K * LS];
    SauvolaJob job = MULTI ? jobs[blockIdx.z] : job1;
    const int X0 = blockIdx.x * P.two;
    const int Y0 = blockIdx.y * P.th;
    if (X0 >= job.w || Y0 >= job.h) return;
    sauvola_tile<K, false, BOTH, PL>(job, P, lds_offset(EBuf), 0u, 0u, X0, Y0, (int)threadIdx.x);
}

// table kernel: NW waves per workgroup share one LDS copy of the decision table (and of the border records); each
// wave then works on its own tile exactly like the general kernel's single wave (no barrier after the staging).
// Tiles of a job are numbered strip-major; workgroup b's wave v takes tile b * NW + v.
template <int K, bool MULTI, bool BOTH, int PL, int NW, int WPE, int NST = 0>
__global__ __launch_bounds__(64 * NW, WPE) void sauvola_tab_kernel(SauvolaJob job1, const SauvolaJob *jobs,
                                                                   SauvolaParams P) {
    constexpr int LS = 64 + 2 * PL;
    __shared__ uint2 EBuf[NW * K * LS];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    {
        typedef const u32x4 __attribute__((address_space(1))) *gc_u4p;
        const gc_u4p src = (gc_u4p)(uintptr_t)P.tab;
        u32x4 *dst = (u32x4 *)dyn_lds;
        const int n16 = P.tab_bytes >> 4;
        for (int i = threadIdx.x; i < n16; i += 64 * NW) dst[i] = src[i];
        const gc_u4p rsrc = (gc_u4p)(uintptr_t)P.colrec;
        u32x4 *rdst = (u32x4 *)(dyn_lds + P.tab_bytes);
        for (int i = threadIdx.x; i < P.colrec_n; i += 64 * NW) rdst[i] = rsrc[i];
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int tile = blockIdx.x * NW + wave;
    const int tx = tile % P.strips, ty = tile / P.strips;
    SauvolaJob job = MULTI ? jobs[blockIdx.z] : job1;
    const int X0 = tx * P.two;
    const int Y0 = ty * P.th;
    if (ty >= P.ytiles || X0 >= job.w || Y0 >= job.h) return;
    const unsigned tab_lds = lds_offset(dyn_lds);
    sauvola_tile<K, true, BOTH, PL, NST>(job, P, lds_offset(EBuf) + (unsigned)(wave * K * LS * 8), tab_lds,
                                    tab_lds + (unsigned)P.tab_bytes, X0, Y0, lane);
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

// Count tables are a function of the window only: built once per (ww, wh) and kept for the life of the process
// (a few hundred bytes each): per row count nrows (count = ww * nrows) and, for the full window height, per column
// count ncols (count = ncols * wh: strips at the left / right image border).
struct SauvolaCounts {
    int dev, ww, wh;
    SauvolaRow *d_rows = nullptr;
    uint4 *d_cols = nullptr;       // [ww + 1] {ms, ss, count, ok}
    int cols_ok = 0;
};
static const SauvolaCounts *count_tables(mrchip_ctx *ctx, int ww, int wh) {
    static std::vector<SauvolaCounts *> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (auto *e : cache)
        if (e->dev == ctx->device && e->ww == ww && e->wh == wh) return e;
    auto *e = new SauvolaCounts;
    e->dev = ctx->device; e->ww = ww; e->wh = wh;
    std::vector<SauvolaRow> rows(wh + 1, SauvolaRow{});
    for (int nr = 1; nr <= wh; nr++) {
        SauvolaRow &r = rows[nr];
        const unsigned long long c = (unsigned long long)ww * nr;
        r.c255 = (unsigned)(255ull * c); r.c65025 = (unsigned)(65025ull * c);
        r.ok = 65026ull * c <= 0xffffffffull && magic_for(c, 255ull * c, &r.ms, &r.ss);
    }
    std::vector<uint4> cols(ww + 1, uint4{0, 0, 1, 0});
    e->cols_ok = 1;
    for (int nc = 1; nc <= ww; nc++) {
        const unsigned long long c = (unsigned long long)nc * wh;
        unsigned ms = 0; int ss = 0;
        const bool ok = 65026ull * c <= 0xffffffffull && magic_for(c, 255ull * c, &ms, &ss);
        cols[nc] = uint4{ms, (unsigned)ss, (unsigned)c, ok ? 1u : 0u};
        if (!ok) e->cols_ok = 0;
    }
    // (uploads through the library's page-locked staging: ctx.hip, "copies between the device and ordinary host memory")
    if (hipMalloc((void **)&e->d_rows, rows.size() * sizeof(SauvolaRow)) != hipSuccess ||
        hipMalloc((void **)&e->d_cols, cols.size() * sizeof(uint4)) != hipSuccess ||
        upload_1d(ctx->streams[0], e->d_rows, rows.data(), rows.size() * sizeof(SauvolaRow)) != 0 ||
        upload_1d(ctx->streams[0], e->d_cols, cols.data(), cols.size() * sizeof(uint4)) != 0 ||
        hipStreamSynchronize(ctx->streams[0]) != hipSuccess) {
        delete e;
        return nullptr;
    }
    cache.push_back(e);
    return e;
}

// The decision table of one (k, R): built on the device on first use (bisection with the general path's predicate),
// compacted on the host to the band of d = px - mean whose entries are not constant, kept for the life of the process
// (a bounded number of them).
struct SauvolaTable {
    int dev; double k, R;
    unsigned short *d_tab = nullptr;
    int W = 0, dlo1 = 0, dhi1 = 0, bytes = 0;
    bool ok = false;
};
static const SauvolaTable *decision_table(mrchip_ctx *ctx, double k, double R) {
    static std::vector<SauvolaTable *> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (auto *e : cache)
        if (e->dev == ctx->device && e->k == k && e->R == R) return e;
    // The cache is bounded: at most MAX_TABLES tables per process (<= 130 KB of device memory each); a caller that
    // keeps inventing (k, R) pairs beyond that gets the general fp64 path for the new ones -- exact, slower -- instead of
    // an unbounded leak.  Entries are never freed (launches in flight read them without a lock).
    constexpr size_t MAX_TABLES = 64;
    static const SauvolaTable none = {};
    if (cache.size() >= MAX_TABLES) return &none;
    auto *e = new SauvolaTable;
    e->dev = ctx->device; e->k = k; e->R = R;
    cache.push_back(e);                      // a failed build is remembered too (ok = false: the general path)
    if (!(k >= 0)) return e;
    unsigned short *d_full = nullptr;
    std::vector<unsigned short> full(65536);
    hipStream_t st = nullptr;
    bool good = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                hipMalloc((void **)&d_full, 65536 * 2) == hipSuccess;
    if (good) {
        hipLaunchKernelGGL(sauvola_t2_build_kernel, dim3(256), dim3(256), 0, st, d_full, k - 1, k * k / R / R);   // pyx:62
        // (the kernel is waited for BEFORE the copy into pageable memory is handed to the runtime: ctx.hip, download_1d)
        good = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
               download_1d(st, full.data(), d_full, 65536 * 2) == 0 && hipStreamSynchronize(st) == hipSuccess;
    }
    if (d_full) (void)hipFree(d_full);
    if (st) (void)hipStreamDestroy(st);
    if (!good) return e;
    int dlo = 256, dhi = -256;               // d range of the entries that are not "always" / not "never"
    for (int m = 0; m < 256; m++)
        for (int px = 0; px < 256; px++) {
            const unsigned short t = full[m * 256 + px];
            if (t != 0) dlo = std::min(dlo, px - m);
            if (t != 65026) dhi = std::max(dhi, px - m);
        }
    if (dlo > 255 || dhi < -255) return e;   // a constant decision: not worth a table
    e->dlo1 = dlo - 1; e->dhi1 = dhi + 1;
    e->W = e->dhi1 - e->dlo1 + 1;
    std::vector<unsigned short> tab((size_t)256 * e->W + 8, 0);
    for (int m = 0; m < 256; m++)
        for (int j = 0; j < e->W; j++) {
            const int d = e->dlo1 + j, px = m + d;
            unsigned short v = 0;
            if (j == 0) v = 0;                          // every d below the band: tmp <= 0 for every mean
            else if (j == e->W - 1) v = 65026;          // every d above it: never
            else if (px >= 0 && px <= 255) v = full[m * 256 + px];
            tab[(size_t)m * e->W + j] = v;
        }
    e->bytes = (int)((256 * e->W * 2 + 15) & ~15);
    if (hipMalloc((void **)&e->d_tab, e->bytes) != hipSuccess ||
        upload_1d(ctx->streams[0], e->d_tab, tab.data(), e->bytes) != 0 || hipStreamSynchronize(ctx->streams[0]) != hipSuccess)
        return e;
    e->ok = true;
    return e;
}

// every (mean, px, var) the kernel can see, table against predicate (see sauvola_t2_selftest_kernel)
int sauvola_table_selftest(mrchip_ctx *ctx, hipStream_t s, double k, double R, unsigned long long *d_bad_tested, int *table_bytes) {
    const SauvolaTable *t = decision_table(ctx, k, R);
    if (!t || !t->ok) { set_error("sauvola: no decision table for k = %g, R = %g", k, R); return MRCHIP_E_UNSUPPORTED; }
    HIP_TRY(hipMemsetAsync(d_bad_tested, 0, 16, s));
    hipLaunchKernelGGL(sauvola_t2_selftest_kernel, dim3(256), dim3(256), 0, s, t->d_tab, t->W, t->dlo1, t->dhi1, k - 1,
                       k * k / R / R, d_bad_tested, d_bad_tested + 1);
    HIP_TRY(hipGetLastError());
    if (table_bytes) *table_bytes = t->bytes;
    return 0;
}

// opt-in of a kernel (on the current device) to more dynamic LDS than the default limit
template <class F>
static int allow_dynamic_lds(F *kernel, int bytes) {
    HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

template <int K>
static int launch_k(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *h_jobs, const SauvolaJob *d_jobs,
                    int njobs, SauvolaParams P, int maxw, int maxh, double alg_bytes) {
    constexpr int CW = 64 * K;
    P.two = (CW - P.ww - K) & ~3;
    bool want_bits = false;
    for (int i = 0; i < njobs; i++) want_bits = want_bits || h_jobs[i].bits != nullptr;
    // Counted stores (sauvola_tile, NST): the 8-column single-polarity table kernel when every job's rows start at
    // 8-byte boundaries (then a lane's 8 columns are wholly inside or outside a strip's outputs) and its rows have room
    // for the zeros a lane at the right image edge stores past column w; all jobs with the 1-bpp rows or none.
    int nst = 0;
    if (K == 8 && h_jobs[0].dst_inv == nullptr) {
        nst = want_bits ? 2 : 1;
        for (int i = 0; i < njobs; i++) {
            const SauvolaJob &j = h_jobs[i];
            if (((uintptr_t)j.src & 7) || ((uintptr_t)j.dst & 3) || j.dst_inv || j.dst_pitch < round_up(j.w, 8) ||
                (want_bits && !j.bits) || (j.bits && j.bits_pitch < cdiv(j.w, 8)))
                nst = 0;
        }
        if (nst == 2) {                      // every job's bytes unwanted: the bit rows only
            bool nb = true;
            for (int i = 0; i < njobs; i++) nb = nb && h_jobs[i].no_bytes;
            if (nb) nst = 3;
        }
        const char *cs_env = getenv("MRCHIP_SAUVOLA_COUNTED_STORES");      // (read per launch: the parity tests run both)
        if (cs_env && atoi(cs_env) == 0) nst = 0;
        if (nst) P.two &= ~(K - 1);
    }
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
    // Rows per tile.  Tall tiles amortise the (wh-1)-row warm-up, but the launch runs in rounds of `slots` resident waves
    // and a last round that is a third full costs a whole one (128 pages 4000x3000 at 256 rows: 13 824 tiles on 4 096
    // slots = 3.4 rounds paid as 4): take the number of equal tile rows whose rounds x (rows + warm-up) is smallest.
    int strips = cdiv(maxw, P.two);
    int th = 256;
    {
        const long long slots = (long long)(ctx->cus > 0 ? ctx->cus : 256) * (K >= 8 ? 16 : 24);    // resident waves
        double best = 1e30;
        for (int yt = 1; yt <= cdiv(maxh, 32); yt++) {
            const int t = std::max(32, cdiv(maxh, yt));
            if (t > 1024 && yt < cdiv(maxh, 32)) continue;        // very long tiles leave no slack for uneven progress
            const long long tiles = (long long)strips * cdiv(maxh, t) * njobs;
            const double cost = (double)cdiv((int)std::min<long long>(tiles, 1 << 30), (int)slots) * (t + 0.5 * P.wh);
            if (cost < best * 0.999) { best = cost; th = t; }
        }
    }
    P.th = th;
    P.strips = strips; P.ytiles = cdiv(maxh, P.th);
    const char *nm = (njobs == 1 && !d_jobs) ? "sauvola" : (h_jobs[0].dst_inv ? "sauvola_boxes" : "sauvola");
    const bool single = njobs == 1 && !d_jobs;
    const bool both = h_jobs[0].dst_inv != nullptr;
    for (int i = 1; i < njobs; i++)
        if ((h_jobs[i].dst_inv != nullptr) != both) { set_error("sauvola: jobs with and without a second polarity in one launch"); return MRCHIP_E_ARG; }
    // slack lanes of the LDS prefix rows: 8 when K * 8 columns cover half the window (the pipeline's windows), else 32
    const bool small_pl = K <= 8 && P.l + K <= K * 8;
    const int sel = (single ? 0 : 2) | (both ? 1 : 0);
    if constexpr (K <= 8) {
        // table kernel: NW waves share one LDS copy of the table -- 16 waves (a CU's worth at 4 per SIMD) for the
        // 8-column page kernel, 8 for the 4-column kernel of the hOCR boxes (several workgroups per CU)
        constexpr int NW = K == 8 ? 16 : 8, WPE = K == 8 ? 4 : 6;
        const int dyn = P.tab_bytes + 16 * P.colrec_n;
        const int stat = NW * K * (64 + 2 * (small_pl ? 8 : 32)) * 8;
        // (K = 8 with two polarities never gets here with a table: sauvola_columns_per_lane keeps box launches of table-sized
        // windows on 4 columns, and wider windows have no small_pl)
        if (P.tab && (K == 4 || (small_pl && !both)) && stat + dyn <= 160 * 1024) {
            dim3 grid(cdiv(P.strips * P.ytiles, NW), 1, njobs);
#define SAUVOLA_TAB_LAUNCH(MULTI_, BOTH_, PL_) SAUVOLA_TAB_LAUNCH_N(MULTI_, BOTH_, PL_, 0)
#define SAUVOLA_TAB_LAUNCH_N(MULTI_, BOTH_, PL_, NST_)                                                                 \
    do {                                                                                                               \
        auto *kern = sauvola_tab_kernel<K, MULTI_, BOTH_, PL_, NW, WPE, NST_>;                                         \
        TRY(allow_dynamic_lds(kern, dyn));        /* per device, cheap: set on every launch (several GPUs per process) */ \
        LAUNCH(ctx, s, nm, alg_bytes, hipLaunchKernelGGL(kern, grid, dim3(64 * NW), dyn, s, h_jobs[0], d_jobs, P));    \
    } while (0)
#define SAUVOLA_TAB_LAUNCH_PL(MULTI_, BOTH_)                                                                           \
    do {                                                                                                               \
        if (small_pl) SAUVOLA_TAB_LAUNCH(MULTI_, BOTH_, 8);                                                            \
        else if constexpr (K == 4) SAUVOLA_TAB_LAUNCH(MULTI_, BOTH_, 32);                                              \
    } while (0)
            if constexpr (K == 8) {
                if (nst && small_pl) {         // (sel is 0 or 2 here: single polarity)
                    if (sel == 0) {
                        if (nst == 3) SAUVOLA_TAB_LAUNCH_N(false, false, 8, 3);
                        else if (nst == 2) SAUVOLA_TAB_LAUNCH_N(false, false, 8, 2);
                        else SAUVOLA_TAB_LAUNCH_N(false, false, 8, 1);
                    } else {
                        if (nst == 3) SAUVOLA_TAB_LAUNCH_N(true, false, 8, 3);
                        else if (nst == 2) SAUVOLA_TAB_LAUNCH_N(true, false, 8, 2);
                        else SAUVOLA_TAB_LAUNCH_N(true, false, 8, 1);
                    }
                    return 0;
                }
            }
            switch (sel) {
                case 0: SAUVOLA_TAB_LAUNCH_PL(false, false); break;
                case 2: SAUVOLA_TAB_LAUNCH_PL(true, false); break;
                case 1: if constexpr (K == 4) SAUVOLA_TAB_LAUNCH_PL(false, true); break;
                default: if constexpr (K == 4) SAUVOLA_TAB_LAUNCH_PL(true, true); break;
            }
#undef SAUVOLA_TAB_LAUNCH_PL
#undef SAUVOLA_TAB_LAUNCH_N
#undef SAUVOLA_TAB_LAUNCH
            return 0;
        }
    }
    dim3 grid(strips, cdiv(maxh, P.th), njobs);
#define SAUVOLA_LAUNCH(MULTI_, BOTH_)                                                                            \
    do {                                                                                                         \
        if (small_pl && K <= 8)                                                                                  \
            LAUNCH(ctx, s, nm, alg_bytes, hipLaunchKernelGGL((sauvola_kernel<K, MULTI_, BOTH_, (K <= 8 ? 8 : 32)>), grid, \
                                                             dim3(64), 0, s, h_jobs[0], d_jobs, P));              \
        else                                                                                                     \
            LAUNCH(ctx, s, nm, alg_bytes, hipLaunchKernelGGL((sauvola_kernel<K, MULTI_, BOTH_, 32>), grid, dim3(64), 0, s, \
                                                             h_jobs[0], d_jobs, P));                              \
    } while (0)
    switch (sel) {
        case 0: SAUVOLA_LAUNCH(false, false); break;
        case 1: SAUVOLA_LAUNCH(false, true); break;
        case 2: SAUVOLA_LAUNCH(true, false); break;
        default: SAUVOLA_LAUNCH(true, true); break;
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
static int sauvola_columns_per_lane(int maxw, int maxh, int ww, bool both) {
    // 8 columns per lane halve the strip halo (452 of 512 columns are outputs instead of 200 of 256) at the
    // price of 128 VGPRs: measured 12 % faster on whole pages, 10 % slower on the short hOCR-box crops.  A two-polarity
    // launch (hOCR boxes) stays on 4 columns whatever the size of its boxes: with the second polarity's table entries
    // and form bits live across the compare loop the 8-column table kernel does not fit 128 VGPRs (8 spilled, and it
    // measured 1.9x slower on boxes), so that instantiation does not exist.
    const bool page_like = maxw >= 1024 && maxh >= 256 && !both;
    if (ww <= 120 && !page_like) return 4;
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
    SauvolaParams P = {};
    P.ww = ww; P.wh = wh;
    P.l = (ww + 1) / 2; P.r = ww / 2; P.o = (wh + 1) / 2; P.u = wh / 2;
    P.k = k; P.km1 = k - 1; P.k2 = k * k / R / R;     // pyx:62
    P.flags = flags;
    const SauvolaCounts *ct = count_tables(ctx, ww, wh);
    if (!ct) { set_error("sauvola: cannot allocate the count tables"); return MRCHIP_E_NOMEM; }
    P.rows = ct->d_rows;
    // Table-driven decision (sauvola_tab_kernel) for k >= 0 wherever the count has a magic number; everything else --
    // k < 0, a (k, R) whose band does not fit the LDS, windows too wide for the 4- / 8-column strips -- and the whole
    // launch under MRCHIP_SAUVOLA_FAST=0 (the parity tests run both) takes the reference's fp64 sequence.
    const char *fast_env = getenv("MRCHIP_SAUVOLA_FAST");
    if (!(fast_env && atoi(fast_env) == 0) && k >= 0) {
        const SauvolaTable *t = decision_table(ctx, k, R);
        if (t && t->ok) {
            P.tab = t->d_tab; P.tab_bytes = t->bytes; P.tabW = t->W; P.dlo1 = t->dlo1; P.dhi1 = t->dhi1;
            P.colrec = ct->d_cols; P.colrec_n = ww + 1; P.colrec_ok = ct->cols_ok;
        }
    }
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
    bool both = false;
    for (int i = 0; i < njobs; i++) both = both || jobs[i].dst_inv != nullptr;
    const int K = sauvola_columns_per_lane(maxw, maxh, ww, both);
    if (K == 4) return launch_k<4>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
    if (K == 8) return launch_k<8>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
    return launch_k<16>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
}


bool sauvola_writes_bits(int maxw, int maxh, int ww) { return sauvola_columns_per_lane(maxw, maxh, ww, false) >= 8; }

int launch_sauvola(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *jobs, int njobs,
                   int ww, int wh, double k, double R, int flags) {
    if (njobs == 1) return launch_sauvola_dev(ctx, s, jobs, nullptr, 1, ww, wh, k, R, flags);
    set_error("launch_sauvola: multi-job launches go through launch_sauvola_dev");
    return MRCHIP_E_ARG;
}

}  // namespace mrchip
