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

#include "mrchip_internal.h"

namespace mrchip {

struct SauvolaParams {
    int ww, wh;       // window
    int l, r, o, u;   // l=(ww+1)/2 r=ww/2 o=(wh+1)/2 u=wh/2  (pyx:76-79)
    double k, km1, k2;
    int flags;
    int two;          // output columns per tile
    int th;           // output rows per tile
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

// Per wave: a strip of CW = 64*K input columns by `rows` output rows (tall tiles amortise the
// (wh-1)-row warm-up).  (An LDS ring of the last wh rows was tried: it removes the re-reads but
// caps the CU at ~10 waves and ran 2x slower -- occupancy is what hides this kernel's per-row chain.)
template <int K, bool MULTI>
__global__ __launch_bounds__(64) void sauvola_kernel(SauvolaJob job1, const SauvolaJob *jobs,
                                                     SauvolaParams P) {
    constexpr int KD = K / 4;
    constexpr int PF = (K == 8) ? 2 : 4;               // rows in flight
    // Prefix rows in LDS, transposed: strip column ci = K*t + i lives at [i][t + PL].  A wave's
    // accesses for one pixel index i are then consecutive dwords (conflict-free); the natural
    // [ci] order would put lanes 16 B apart = a 4-way bank conflict on every read.  PL lanes of
    // slack on both sides: halo lanes evaluate the formula on out-of-strip indices instead of
    // branching around it (their results are never stored).
    constexpr int PL = 32, LS = 64 + 2 * PL;
    __shared__ unsigned EsBuf[K * LS];
    __shared__ unsigned EqBuf[K * LS];
    auto pidx = [&](int ci) { const int c2 = ci + K * PL; return (c2 % K) * LS + c2 / K; };

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
    // warm-up: rows [Y0-o, Y0+u-1] (clipped) enter the sums
    for (int yy = max(0, Y0 - o); yy < min(h, Y0 + u); yy++) {
        unsigned wv[KD];
        gload(yy, wv);
        acc_row_m(wv, true);
    }
    // three register queues, PF rows deep: entering rows y+u, leaving rows y-o, centre rows y.
    // Every address is known in advance, so the loads run PF rows ahead of their use and the
    // serial chain of a row never waits for memory (leaving / centre rows come back from L2/MALL).
    unsigned qe[PF][KD], ql[PF][KD], qc[PF][KD];
    const bool invert = (P.flags & SAUVOLA_INVERT) != 0;
#pragma unroll
    for (int d = 0; d < PF; d++) {
        gload(Y0 + u + d, qe[d]);
        gload(Y0 - o + d, ql[d]);
        gload(Y0 + d, qc[d]);
    }

    unsigned ones_a = 0, ones_b = 0;
    const bool kpos = P.k >= 0;

    for (int y = Y0; y < Y0 + rows; y++) {
        // ---- heads of the queues, then refill PF rows ahead ----
        unsigned ev[KD], lv[KD], cv[KD];
#pragma unroll
        for (int q = 0; q < KD; q++) { ev[q] = qe[0][q]; lv[q] = ql[0][q]; cv[q] = qc[0][q]; }
#pragma unroll
        for (int d = 0; d + 1 < PF; d++)
#pragma unroll
            for (int q = 0; q < KD; q++) {
                qe[d][q] = qe[d + 1][q]; ql[d][q] = ql[d + 1][q]; qc[d][q] = qc[d + 1][q];
            }
        gload(y + u + PF, qe[PF - 1]);
        gload(y - o + PF, ql[PF - 1]);
        gload(y + PF, qc[PF - 1]);
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
        const int nrows = min(y + u, h - 1) - max(y - o, -1);

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
            EsBuf[i * LS + lane + PL] = bs + ps[i];
            EqBuf[i * LS + lane + PL] = bq + pqx[i];
        }
        lds_wave_sync();

        // wave-uniform count / reciprocal when every output of the strip has the full window width
        const unsigned ucount = (unsigned)(P.ww * nrows);
        const double urcd = rcp_nr((double)ucount), uhrcd = 0.5 * urcd;

        unsigned outa[KD], outb[KD];
#pragma unroll
        for (int q = 0; q < KD; q++) { outa[q] = 0; outb[q] = 0; }
        bool any = false, all = true;
#pragma unroll
        for (int i = 0; i < K; i++) {
            const int c = c0 + i;
            const bool valid = (c >= X0) && (c < X0 + nout);
            any |= valid;
            all &= valid;
            // evaluated for every column of the lane, valid or not (no divergent branch per pixel)
            const int ci = K * lane + i;
            const int ia = pidx(ci + r + 1), ib = pidx(ci - l + 1);
            const unsigned S = EsBuf[ia] - EsBuf[ib];
            const unsigned Q = EqBuf[ia] - EqBuf[ib];
            unsigned count = ucount;
            double rcd = urcd, hrcd = uhrcd;
            if (!full_cols) {
                const int ncols = min(c + r, w - 1) - max(c - l + 1, 0) + 1;
                count = (unsigned)max(ncols * nrows, 1);
                rcd = rcp_nr((double)count);
                hrcd = 0.5 * rcd;
            }
            const unsigned px = (cv[i / 4] >> (8 * (i & 3))) & 0xffu;     // only valid columns are stored
            const double Sd = (double)S, Qd = (double)Q, pxd = (double)px;
            const bool form = sauvola_form_dd(Sd, Qd, pxd, rcd, hrcd, kpos, P.km1, P.k2);   // pyx:144-151
            const unsigned bit = valid ? ((form ? 0u : 1u) ^ (invert ? 1u : 0u)) : 0u;   // pyx:153 (+ mrc.py:85)
            outa[i / 4] |= bit << (8 * (i & 3));
            ones_a += bit;
            if (job.dst_inv) {
                // the same window on the image 255-p (mrc.py:224, 235)
                // sum(255-p) = 255 n - S, sum((255-p)^2) = 65025 n - 510 S + Q: integers below 2^32, exact in
                // fp64 whatever the rounding of the fmas (three conversions saved)
                const double cd = (double)count;
                const double Sid = __builtin_fma(255.0, cd, -Sd);
                const double Qid = __builtin_fma(-510.0, Sd, __builtin_fma(65025.0, cd, Qd));
                const bool fi = sauvola_form_dd(Sid, Qid, 255.0 - pxd, rcd, hrcd, kpos, P.km1, P.k2);
                const unsigned bi = valid ? ((fi ? 0u : 1u) ^ (invert ? 1u : 0u)) : 0u;
                outb[i / 4] |= bi << (8 * (i & 3));
                ones_b += bi;
            }
        }
        if (any) {
            uint8_t *d = job.dst + (size_t)y * job.dst_pitch + c0;
            const bool aligned = (reinterpret_cast<uintptr_t>(d) & 3u) == 0;
            if (all && aligned) {
#pragma unroll
                for (int q = 0; q < KD; q++) reinterpret_cast<unsigned *>(d)[q] = outa[q];
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
            if (job.dst_inv) {
                uint8_t *e = job.dst_inv + (size_t)y * job.dst_pitch + c0;
                if (all && aligned) {
#pragma unroll
                    for (int q = 0; q < KD; q++) reinterpret_cast<unsigned *>(e)[q] = outb[q];
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
    }
    if (job.counts) {
        unsigned a = wave_sum(ones_a);
        unsigned b = wave_sum(ones_b);
        if (lane == 0) {
            if (a) atomicAdd(&job.counts[0], a);
            if (b) atomicAdd(&job.counts[1], b);
        }
    }
}

template <int K>
static int launch_k(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *h_jobs, const SauvolaJob *d_jobs,
                    int njobs, SauvolaParams P, int maxw, int maxh, double alg_bytes) {
    constexpr int CW = 64 * K;
    P.two = (CW - P.ww - K) & ~3;
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
    if (njobs == 1 && !d_jobs)
        LAUNCH(ctx, s, nm, alg_bytes,
               hipLaunchKernelGGL((sauvola_kernel<K, false>), grid, dim3(64), 0, s, h_jobs[0], d_jobs, P));
    else
        LAUNCH(ctx, s, nm, alg_bytes,
               hipLaunchKernelGGL((sauvola_kernel<K, true>), grid, dim3(64), 0, s, h_jobs[0], d_jobs, P));
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
    static const int force_k = getenv("MRCHIP_SAUVOLA_K") ? atoi(getenv("MRCHIP_SAUVOLA_K")) : 0;   // tuning knob
    // 8 columns per lane halve the strip halo (452 of 512 columns are outputs instead of 200 of 256) at the
    // price of 128 VGPRs: measured 12 % faster on whole pages, 10 % slower on the short hOCR-box crops
    const bool page_like = maxw >= 1024 && maxh >= 256;
    if (ww <= 120 && force_k != 8 && !(page_like && force_k != 4))
        return launch_k<4>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
    if (ww <= 360) return launch_k<8>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
    return launch_k<16>(ctx, s, jobs, d_jobs, njobs, P, maxw, maxh, alg);
}

int launch_sauvola(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *jobs, int njobs,
                   int ww, int wh, double k, double R, int flags) {
    if (njobs == 1) return launch_sauvola_dev(ctx, s, jobs, nullptr, 1, ww, wh, k, R, flags);
    set_error("launch_sauvola: multi-job launches go through launch_sauvola_dev");
    return MRCHIP_E_ARG;
}

}  // namespace mrchip
