// fast_mask_denoise (reference: cython/optimiser.pyx:436-472) for gfx950.
//
// The reference sweeps the mask in raster order IN PLACE: a set pixel survives
// iff at least `mincnt` OTHER pixels of its (2n+1)^2 window are set, where the
// neighbours earlier in raster order are already updated.  The result is the
// unique solution f of
//     f(p) = m(p) & ( sum_{q earlier} f(q) + sum_{q later} m(q) >= mincnt )
// on the inner rectangle [n,H-n) x [n,W-n); everything else keeps m.
//
// Fast path (n=2, mincnt=4 -- the only call site, mrc.py:388): bit-packed and
// bit-sliced.  pack: bytes -> 1 bit/pixel (32 px per dword).  solve: ONE wave
// walks the page top to bottom; each lane holds KW consecutive 32-pixel words of
// a row, so a wave spans 2048*KW columns entirely in registers.  Per row the
// neighbour count is built with carry-save adders on whole words (5-tap
// horizontal sums of the two final rows above and the two original rows below,
// saturating at 4), pixels are classified as keep / drop / "needs one or two of
// its two left neighbours", and the left-to-right recurrence inside the row is
// resolved by a monotone word-parallel iteration (funnel shifts + one lane
// shuffle per sweep) that stops when no bit changes -- a few sweeps on text,
// chain-length sweeps on a 1-px rule.  Rows are inherently sequential (row y
// needs the FINAL rows y-1, y-2), pages are independent: a batch runs one wave
// per page.  unpack: bits -> bytes.
//
// General (n, mincnt): byte-domain Jacobi sweeps to the same unique fixpoint.
//
// Algorithmic bytes: 2*w*h per call.
#include <cstdlib>

#include "mrchip_internal.h"

namespace mrchip {

// ---- pack / unpack ----------------------------------------------------------
__device__ __forceinline__ unsigned nz_nibble(unsigned d) {
    unsigned t = (((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u;   // 0x80 per nonzero byte
    t >>= 7;
    return (t * 0x01020408u) >> 24 & 0xFu;
}

__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *mask, int pitch, size_t mstride, int w, int h,
                                                        unsigned *bits, int wpr /* words per row */, size_t bstride) {
    const int y = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= wpr) return;
    mask += (size_t)blockIdx.z * mstride;
    bits += (size_t)blockIdx.z * bstride;
    const uint8_t *row = mask + (size_t)y * pitch + (size_t)j * 32;
    unsigned word = 0;
    if (j * 32 + 32 <= w) {
        const uint4 *p = reinterpret_cast<const uint4 *>(row);
        uint4 a = p[0], b = p[1];
        word = nz_nibble(a.x) | nz_nibble(a.y) << 4 | nz_nibble(a.z) << 8 | nz_nibble(a.w) << 12 |
               nz_nibble(b.x) << 16 | nz_nibble(b.y) << 20 | nz_nibble(b.z) << 24 | nz_nibble(b.w) << 28;
    } else {
        for (int i = 0; j * 32 + i < w; i++) word |= (row[i] ? 1u : 0u) << i;
    }
    bits[(size_t)y * wpr + j] = word;
}

int launch_pack_bits(mrchip_ctx *ctx, hipStream_t s, const uint8_t *mask, int pitch, int w, int h, unsigned *bits, int wpr) {
    if (w <= 0 || h <= 0) return 0;
    if (h > MAX_GRID_Z) { set_error("pack_bits: %d rows", h); return MRCHIP_E_UNSUPPORTED; }
    LAUNCH(ctx, s, "mask_pack_bits", 1.125 * w * h,
           hipLaunchKernelGGL(pack_bits_kernel, dim3(cdiv(wpr, 256), h, 1), dim3(256), 0, s, mask, pitch, (size_t)0, w, h, bits,
                              wpr, (size_t)0));
    return 0;
}

// 1 bit per pixel, most significant bit first, rows padded to whole bytes: the layout of a raw PBM
// (P4) body and of PIL's mode '1' -- what mrc.encode_mrc_mask (mrc.py:474-520) turns the bool mask
// into before it goes to jbig2/PNG.  One thread per output byte (8 mask bytes in).
__device__ __forceinline__ unsigned nz_nibble_msb(unsigned m) {
    unsigned t = (((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m) & 0x80808080u;
    return (((t >> 7) * 0x08040201u) >> 24) & 0xFu;       // byte 0 -> bit 3 ... byte 3 -> bit 0
}
__global__ __launch_bounds__(256) void pack_msb_kernel(const uint8_t *mask, int pitch, size_t mstride, int w, int h,
                                                       uint8_t *out, int bpr /* bytes per row */, size_t ostride) {
    const int y = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= bpr) return;
    mask += (size_t)blockIdx.z * mstride;
    out += (size_t)blockIdx.z * ostride;
    const uint8_t *row = mask + (size_t)y * pitch + (size_t)j * 8;
    unsigned v;
    if (j * 8 + 8 <= w) {
        const uint2 a = *reinterpret_cast<const uint2 *>(row);
        v = nz_nibble_msb(a.x) << 4 | nz_nibble_msb(a.y);
    } else {
        v = 0;
        for (int i = 0; j * 8 + i < w; i++) v |= (row[i] ? 0x80u : 0u) >> i;
    }
    out[(size_t)y * bpr + j] = (uint8_t)v;
}

int launch_pack_msb(mrchip_ctx *ctx, hipStream_t s, Plane mask, int w, int h, uint8_t *out, size_t ostride, int npages) {
    const int bpr = (w + 7) / 8;
    LAUNCH(ctx, s, "mask_pack_1bpp", 1.125 * w * h * npages,
           hipLaunchKernelGGL(pack_msb_kernel, dim3(cdiv(bpr, 256), h, npages), dim3(256), 0, s, mask.p, mask.pitch, mask.stride,
                              w, h, out, bpr, ostride));
    return 0;
}

__device__ __forceinline__ unsigned spread_nibble(unsigned nib) { return (nib * 0x00204081u) & 0x01010101u; }

// rowflags (optional): byte y of a page = 1 when row y of the finished mask has a set pixel (pre-zeroed by the launcher);
// the band walkers of optimise cut the bg layer at the rows without ink (k_optimise.hip, OptBand)
__global__ __launch_bounds__(256) void unpack_bits_kernel(const unsigned *bits, int wpr, size_t bstride, uint8_t *mask,
                                                          int pitch, size_t mstride, int w, int h, uint8_t *rowflags,
                                                          size_t rfstride) {
    const int y = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    unsigned word = 0;
    if (j < wpr) {
        mask += (size_t)blockIdx.z * mstride;
        bits += (size_t)blockIdx.z * bstride;
        word = bits[(size_t)y * wpr + j];
        if (j == wpr - 1 && (w & 31)) word &= (1u << (w & 31)) - 1u;
        uint8_t *row = mask + (size_t)y * pitch + (size_t)j * 32;
        if (j * 32 + 32 <= w) {
            uint4 a, b;
            a.x = spread_nibble(word & 0xF); a.y = spread_nibble((word >> 4) & 0xF);
            a.z = spread_nibble((word >> 8) & 0xF); a.w = spread_nibble((word >> 12) & 0xF);
            b.x = spread_nibble((word >> 16) & 0xF); b.y = spread_nibble((word >> 20) & 0xF);
            b.z = spread_nibble((word >> 24) & 0xF); b.w = spread_nibble(word >> 28);
            // non-temporal: the bytes are the pipeline's output and are not read again by any kernel -- written around the L2
            // they neither wait for a line fill nor evict the rows other kernels re-read (0.45 -> 0.35 ms per 128 pages)
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u4{a.x, a.y, a.z, a.w}, reinterpret_cast<u4 *>(row));
            __builtin_nontemporal_store(u4{b.x, b.y, b.z, b.w}, reinterpret_cast<u4 *>(row) + 1);
        } else {
            for (int i = 0; j * 32 + i < w; i++) row[i] = (word >> i) & 1u;
        }
    }
    if (rowflags && __any(word != 0u) && (threadIdx.x & 63) == 0)         // one store per wave with a set pixel (same value: no race)
        rowflags[(size_t)blockIdx.z * rfstride + y] = 1;
}

// ---- bit-sliced sequential solve (n = 2, mincnt = 4) --------------------------
// saturating 3-plane counter: value = b0 + 2*b1, hi = "value >= 4" (sticky)
struct Sat { unsigned b0, b1, hi; };

__device__ __forceinline__ Sat sat_add(const Sat &a, const Sat &b) {
    Sat r;
    unsigned k0 = a.b0 & b.b0;
    r.b0 = a.b0 ^ b.b0;
    unsigned x1 = a.b1 ^ b.b1;
    r.b1 = x1 ^ k0;
    unsigned k1 = (a.b1 & b.b1) | (k0 & x1);
    r.hi = a.hi | b.hi | k1;
    return r;
}

// value at pixel x+d brought to bit position x; L = previous word (lower x), R = next word
__device__ __forceinline__ unsigned sh_p1(unsigned cur, unsigned R) { return __builtin_amdgcn_alignbit(R, cur, 1); }
__device__ __forceinline__ unsigned sh_p2(unsigned cur, unsigned R) { return __builtin_amdgcn_alignbit(R, cur, 2); }
__device__ __forceinline__ unsigned sh_m1(unsigned cur, unsigned L) { return __builtin_amdgcn_alignbit(cur, L, 31); }
__device__ __forceinline__ unsigned sh_m2(unsigned cur, unsigned L) { return __builtin_amdgcn_alignbit(cur, L, 30); }

template <int KW>
struct Row { unsigned w[KW]; };

template <int KW>
__device__ __forceinline__ void neighbours(const Row<KW> &r, int lane, unsigned (&L)[KW], unsigned (&R)[KW]) {
    unsigned up = __shfl_up(r.w[KW - 1], 1);      // last word of the previous lane
    unsigned dn = __shfl_down(r.w[0], 1);         // first word of the next lane
    if (lane == 0) up = 0;
    if (lane == 63) dn = 0;
#pragma unroll
    for (int k = 0; k < KW; k++) {
        L[k] = k > 0 ? r.w[k - 1] : up;
        R[k] = k < KW - 1 ? r.w[k + 1] : dn;
    }
}

// 5-tap horizontal sum (x-2..x+2) of a bit row -> Sat (0..5: b0, b1, hi=fours)
template <int KW>
__device__ __forceinline__ void h5(const Row<KW> &r, int lane, Sat (&out)[KW]) {
    unsigned L[KW], R[KW];
    neighbours<KW>(r, lane, L, R);
#pragma unroll
    for (int k = 0; k < KW; k++) {
        unsigned c = r.w[k];
        unsigned a = sh_m2(c, L[k]), b = sh_m1(c, L[k]), d = sh_p1(c, R[k]), e = sh_p2(c, R[k]);
        unsigned ab = a ^ b;
        unsigned s1 = ab ^ c;
        unsigned c1 = (a & b) | (c & ab);
        unsigned s2 = d ^ e, c2 = d & e;
        unsigned c3 = s1 & s2;
        out[k].b0 = s1 ^ s2;
        unsigned t = c1 ^ c2;
        out[k].b1 = t ^ c3;
        out[k].hi = (c1 & c2) | (c3 & t);
    }
}

template <int KW>
__device__ __forceinline__ Row<KW> load_row(const unsigned *bits, int wpr, int y, int h, int lane) {
    Row<KW> r;
#pragma unroll
    for (int k = 0; k < KW; k++) {
        int j = lane * KW + k;
        r.w[k] = (y >= 0 && y < h && j < wpr) ? bits[(size_t)y * wpr + j] : 0u;
    }
    return r;
}

// ---- the same solve, bands of rows in parallel ------------------------------------------------------------
// Rows are sequential because row y needs the FINAL rows y-1, y-2.  But the final rows differ from the original
// ones only where a pixel was dropped, and a dropped pixel matters to the rows below only if a pixel there sits on
// the threshold: a band of rows solved from the ORIGINAL two rows above it is almost always right, and where it
// is not, the error dies out after a few rows.  So: (1) every band of DN_BAND rows is solved by its own wave from
// the originals above it (out of place: originals in `org`, results in `fin`); (2) one wave per page walks the band
// boundaries top to bottom: where the true final rows above a band differ from the originals the band assumed, it
// re-solves the band's rows from the true ones until two consecutive rows come out as they already were -- from
// there on everything is identical, because a row depends on nothing but the two finals above it and originals.
// The result is exactly the sequential one (the reconciliation is itself sequential, just short).
constexpr int DN_BAND = 96;

template <int KW>
struct RowSolver {
    const unsigned *org;
    int wpr, h, lane;
    unsigned inner[KW];
    Row<KW> m0, m1, m2;
    Sat hf2[KW], hf1[KW], hm1[KW], hm2[KW];

    __device__ __forceinline__ void init(const unsigned *org_, int wpr_, int w, int h_, int lane_) {
        org = org_; wpr = wpr_; h = h_; lane = lane_;
#pragma unroll
        for (int k = 0; k < KW; k++) {
            const int x0 = (lane * KW + k) * 32;
            unsigned m = 0;
            for (int i = 0; i < 32; i++) {
                const int x = x0 + i;
                if (x >= 2 && x < w - 2) m |= 1u << i;
            }
            inner[k] = m;
        }
    }
    // ready to solve row y: f2 / f1 = final rows y-2, y-1
    __device__ __forceinline__ void start(int y, const Row<KW> &f2, const Row<KW> &f1) {
        m0 = load_row<KW>(org, wpr, y, h, lane);
        m1 = load_row<KW>(org, wpr, y + 1, h, lane);
        m2 = load_row<KW>(org, wpr, y + 2, h, lane);
        h5<KW>(f2, lane, hf2);
        h5<KW>(f1, lane, hf1);
        h5<KW>(m1, lane, hm1);
        h5<KW>(m2, lane, hm2);
    }
    // final row y (the row `start` / the previous `step` prepared), then advance to row y + 1; `next2` = original row y + 3
    __device__ __forceinline__ Row<KW> step(const Row<KW> &next2) {
        unsigned always[KW], t1[KW], t2[KW], fixed[KW], upd[KW];
        {
            unsigned L[KW], R[KW];
            neighbours<KW>(m0, lane, L, R);
#pragma unroll
            for (int k = 0; k < KW; k++) {
                Sat s = sat_add(sat_add(hf2[k], hf1[k]), sat_add(hm1[k], hm2[k]));
                unsigned d = sh_p1(m0.w[k], R[k]), e = sh_p2(m0.w[k], R[k]);
                Sat rr; rr.b0 = d ^ e; rr.b1 = d & e; rr.hi = 0;
                s = sat_add(s, rr);
                always[k] = s.hi;
                t1[k] = ~s.hi & s.b1 & s.b0;
                t2[k] = ~s.hi & s.b1 & ~s.b0;
                upd[k] = m0.w[k] & inner[k];
                fixed[k] = m0.w[k] & ~inner[k];
            }
        }
        Row<KW> f = m0;
        for (;;) {
            unsigned up = __shfl_up(f.w[KW - 1], 1);
            if (lane == 0) up = 0;
            unsigned changed = 0;
            Row<KW> g;
#pragma unroll
            for (int k = 0; k < KW; k++) {
                unsigned Lw = k > 0 ? f.w[k - 1] : up;
                unsigned a = sh_m1(f.w[k], Lw), b = sh_m2(f.w[k], Lw);
                unsigned nf = fixed[k] | (upd[k] & (always[k] | (t1[k] & (a | b)) | (t2[k] & a & b)));
                changed |= nf ^ f.w[k];
                g.w[k] = nf;
            }
            f = g;
            if (!__any(changed != 0)) break;
        }
#pragma unroll
        for (int k = 0; k < KW; k++) { hf2[k] = hf1[k]; hm1[k] = hm2[k]; }
        h5<KW>(f, lane, hf1);
        m0 = m1; m1 = m2; m2 = next2;
        h5<KW>(m2, lane, hm2);
        return f;
    }
};

template <int KW>
__device__ __forceinline__ void store_row(unsigned *fin, int wpr, int y, int lane, const Row<KW> &f) {
#pragma unroll
    for (int k = 0; k < KW; k++) {
        const int j = lane * KW + k;
        if (j < wpr) fin[(size_t)y * wpr + j] = f.w[k];
    }
}

template <int KW>
__device__ __forceinline__ bool rows_equal(const Row<KW> &a, const Row<KW> &b) {
    unsigned d = 0;
#pragma unroll
    for (int k = 0; k < KW; k++) d |= a.w[k] ^ b.w[k];
    return !__any(d != 0);
}

// grid (bands, pages); per page: fin = bits, org = bits + wpr * h
template <int KW>
__global__ __launch_bounds__(64) void denoise_band_kernel(unsigned *bits, int wpr, size_t bstride, int w, int h) {
    const int lane = threadIdx.x;
    unsigned *fin = bits + (size_t)blockIdx.y * bstride;
    const unsigned *org = fin + (size_t)wpr * h;
    const int y0 = max(2, (int)blockIdx.x * DN_BAND), y1 = min(h - 2, ((int)blockIdx.x + 1) * DN_BAND);
    if (blockIdx.x == 0) {                 // rows outside the inner rectangle keep the original
        store_row<KW>(fin, wpr, 0, lane, load_row<KW>(org, wpr, 0, h, lane));
        store_row<KW>(fin, wpr, 1, lane, load_row<KW>(org, wpr, 1, h, lane));
        store_row<KW>(fin, wpr, h - 2, lane, load_row<KW>(org, wpr, h - 2, h, lane));
        store_row<KW>(fin, wpr, h - 1, lane, load_row<KW>(org, wpr, h - 1, h, lane));
    }
    if (y0 >= y1) return;
    RowSolver<KW> S;
    S.init(org, wpr, w, h, lane);
    // the two rows above: originals (exact for the first band, whose rows 0, 1 are final as they are)
    S.start(y0, load_row<KW>(org, wpr, y0 - 2, h, lane), load_row<KW>(org, wpr, y0 - 1, h, lane));
    constexpr int PF = 4;
    Row<KW> pf[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) pf[i] = load_row<KW>(org, wpr, y0 + 3 + i, h, lane);
    for (int y = y0; y < y1; y++) {
        const Row<KW> f = S.step(pf[0]);
        store_row<KW>(fin, wpr, y, lane, f);
#pragma unroll
        for (int i = 0; i + 1 < PF; i++) pf[i] = pf[i + 1];
        pf[PF - 1] = load_row<KW>(org, wpr, y + 3 + PF, h, lane);
    }
}

// one wave per page: reconcile the band boundaries in order
template <int KW>
__global__ __launch_bounds__(64) void denoise_fix_kernel(unsigned *bits, int wpr, size_t bstride, int w, int h) {
    const int lane = threadIdx.x;
    unsigned *fin = bits + (size_t)blockIdx.x * bstride;
    const unsigned *org = fin + (size_t)wpr * h;
    RowSolver<KW> S;
    S.init(org, wpr, w, h, lane);
    for (int yb = DN_BAND; yb < h - 2; yb += DN_BAND) {
        const Row<KW> t2 = load_row<KW>(fin, wpr, yb - 2, h, lane), t1 = load_row<KW>(fin, wpr, yb - 1, h, lane);
        if (rows_equal<KW>(t2, load_row<KW>(org, wpr, yb - 2, h, lane)) &&
            rows_equal<KW>(t1, load_row<KW>(org, wpr, yb - 1, h, lane)))
            continue;                       // the band below assumed exactly these rows
        S.start(yb, t2, t1);
        int same = 0;
        for (int y = yb; y < h - 2; y++) {
            const Row<KW> f = S.step(load_row<KW>(org, wpr, y + 3, h, lane));
            if (rows_equal<KW>(f, load_row<KW>(fin, wpr, y, h, lane))) {
                if (++same == 2) break;      // two rows in a row as they were: the rest of the page is unchanged
            } else {
                same = 0;
                store_row<KW>(fin, wpr, y, lane, f);
            }
        }
    }
}

// ---- general (n, mincnt): byte-domain Jacobi to the unique fixpoint ------------
__global__ __launch_bounds__(256) void denoise_jacobi_kernel(const uint8_t *orig, const uint8_t *cur, uint8_t *next,
                                                             int pitch, int w, int h, int mincnt, int n,
                                                             int *changed) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const size_t idx = (size_t)y * pitch + x;
    uint8_t o = orig[idx] ? 1 : 0;
    uint8_t v = o;
    if (o && x >= n && x < w - n && y >= n && y < h - n) {
        int cnt = 0;
        for (int dy = -n; dy <= n; dy++)
            for (int dx = -n; dx <= n; dx++) {
                if (dy == 0 && dx == 0) continue;
                const bool earlier = (dy < 0) || (dy == 0 && dx < 0);
                const size_t q = (size_t)(y + dy) * pitch + (x + dx);
                cnt += earlier ? (cur[q] ? 1 : 0) : (orig[q] ? 1 : 0);
            }
        v = cnt >= mincnt ? 1 : 0;
    }
    if (v != (cur[idx] ? 1 : 0)) *changed = 1;
    next[idx] = v;
}

// per page: [final bit rows | original bit rows]; the final rows come first so that `bits + page * stride` is the
// finished mask for the consumers (optimise reads it, unpack turns it into bytes)
template <int KW>
static int launch_seq(mrchip_ctx *ctx, hipStream_t s, unsigned *bits, int wpr, size_t bstride, int w, int h, int npages) {
    LAUNCH(ctx, s, "denoise_solve", 2.0 * w * h / 8 * npages,
           hipLaunchKernelGGL((denoise_band_kernel<KW>), dim3(cdiv(h, DN_BAND), npages), dim3(64), 0, s, bits, wpr, bstride, w, h));
    LAUNCH(ctx, s, "denoise_reconcile", 0.0,
           hipLaunchKernelGGL((denoise_fix_kernel<KW>), dim3(npages), dim3(64), 0, s, bits, wpr, bstride, w, h));
    return 0;
}

// per page: final bit rows | original bit rows | 256 bytes | one flag byte per row ("the finished row has a set pixel")
size_t denoise_rowflags_offset(int w, int h) { return (size_t)cdiv(w, 32) * h * sizeof(unsigned) * 2 + 256; }
size_t denoise_scratch_bytes(int w, int h) { return denoise_rowflags_offset(w, h) + (size_t)round_up(h, 256); }

// bits_ready: the original 1-bpp rows (second half of a page's scratch) already hold the mask -- the Sauvola page
// launch and the hOCR commit wrote them alongside the bytes -- so `pack` is not run
int launch_denoise_batch(mrchip_ctx *ctx, hipStream_t s, Plane mask, int w, int h, int mincnt, int n, unsigned *bits,
                         size_t bits_stride, int npages, bool bits_ready, bool want_rowflags) {
    if (n < 0 || mincnt < 0) { set_error("denoise: negative parameter"); return MRCHIP_E_ARG; }
    if (w <= 2 * n || h <= 2 * n) return 0;      // empty inner rectangle: nothing changes
    const int wpr = cdiv(w, 32);
    const int pitch = mask.pitch;
    if (n == 2 && mincnt == 4 && wpr <= 64 * 8) {
        dim3 grid(cdiv(wpr, 256), h, npages);
        if (!bits_ready)
            LAUNCH(ctx, s, "denoise_pack", 1.0 * w * h * npages,
                   hipLaunchKernelGGL(pack_bits_kernel, grid, dim3(256), 0, s, mask.p, pitch, mask.stride, w, h,
                                      bits + (size_t)wpr * h, wpr, bits_stride));        // originals: second half of a page's scratch
        if (wpr <= 64) TRY(launch_seq<1>(ctx, s, bits, wpr, bits_stride, w, h, npages));
        else if (wpr <= 128) TRY(launch_seq<2>(ctx, s, bits, wpr, bits_stride, w, h, npages));
        else if (wpr <= 256) TRY(launch_seq<4>(ctx, s, bits, wpr, bits_stride, w, h, npages));
        else TRY(launch_seq<8>(ctx, s, bits, wpr, bits_stride, w, h, npages));
        uint8_t *rf = want_rowflags ? reinterpret_cast<uint8_t *>(bits) + denoise_rowflags_offset(w, h) : nullptr;
        if (rf) HIP_TRY(hipMemset2DAsync(rf, bits_stride * 4, 0, (size_t)h, npages, s));
        LAUNCH(ctx, s, "denoise_unpack", 1.0 * w * h * npages,
               hipLaunchKernelGGL(unpack_bits_kernel, grid, dim3(256), 0, s, bits, wpr, bits_stride, mask.p, pitch,
                                  mask.stride, w, h, rf, bits_stride * 4));
        return 0;
    }
  for (int pgi = 0; pgi < npages; pgi++) {
    uint8_t *mask_p = mask.page(pgi);
    // general path
    DevBuf a, b, flag;
    const size_t bytes = (size_t)pitch * h;
    TRY(a.alloc(ctx, bytes));
    TRY(b.alloc(ctx, bytes));
    TRY(flag.alloc(ctx, sizeof(int)));
    HIP_TRY(hipMemcpyAsync(a.p, mask_p, bytes, hipMemcpyDeviceToDevice, s));
    uint8_t *cur = a.as<uint8_t>(), *nxt = b.as<uint8_t>();
    dim3 grid(cdiv(w, 256), h);
    for (long long it = 0; it < (long long)w * h + 2; it++) {
        HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), s));
        LAUNCH(ctx, s, "denoise_jacobi", 2.0 * w * h,
               hipLaunchKernelGGL(denoise_jacobi_kernel, grid, dim3(256), 0, s, mask_p, cur, nxt, pitch, w, h, mincnt, n,
                                  flag.as<int>()));
        uint8_t *t = cur; cur = nxt; nxt = t;
        int changed = 0;
        TRY(download_1d(s, &changed, flag.p, sizeof(int)));
        HIP_TRY(hipStreamSynchronize(s));
        if (!changed) break;
    }
    HIP_TRY(hipMemcpy2DAsync(mask_p, pitch, cur, pitch, w, h, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
    return 0;
}

}  // namespace mrchip
