// PIL Image.thumbnail((int(w/f), int(h/f))) (reference: mrc.py:422-428, 456-462; Pillow
// Image.thumbnail -> Image.reduce (Reduce.c box mean) -> Image.resize BICUBIC (Resample.c,
// 8 bpc fixed point, PRECISION_BITS 22, horizontal pass then vertical pass); SURVEY.md 8a
// row a11).  The size rule and the coefficient tables are host logic (double precision,
// exactly as Pillow computes them); the pixel passes are integer HIP kernels.
// Algorithmic bytes: C*P*(1 + 1/f^2) per layer.
#include <climits>
#include <cmath>
#include <cstring>

#include "mrchip_internal.h"

namespace mrchip {

// ---- host logic ----------------------------------------------------------------
static int round_aspect_w(double number, double aspect, int y) {
    double fl = floor(number), ce = ceil(number);
    double kf = fabs(aspect - fl / y), kc = fabs(aspect - ce / y);
    double pick = (kc < kf) ? ce : fl;          // min() keeps the first on ties
    return pick < 1 ? 1 : (int)pick;
}
static int round_aspect_h(double number, double aspect, int x) {
    double fl = floor(number), ce = ceil(number);
    double kf = fl == 0 ? 0 : fabs(aspect - x / fl), kc = ce == 0 ? 0 : fabs(aspect - x / ce);
    double pick = (kc < kf) ? ce : fl;
    return pick < 1 ? 1 : (int)pick;
}

int thumbnail_size(int w, int h, int req_w, int req_h, int *ow, int *oh) {
    int x = req_w, y = req_h;
    if (x >= w && y >= h) { *ow = w; *oh = h; return 0; }
    double aspect = (double)w / (double)h;
    if ((double)x / (double)y >= aspect) x = round_aspect_w(y * aspect, aspect, y);
    else y = round_aspect_h(x / aspect, aspect, x);
    *ow = x; *oh = y;
    return (x != w || y != h) ? 1 : 0;
}

static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Resample.c sinc_filter / lanczos_filter (support 3): the page-ingest downsample, recode.py:368-372
static double lanczos_filter(double x) {
    auto sinc = [](double v) { if (v == 0.0) return 1.0; v = v * 3.14159265358979323846; return sin(v) / v; };
    if (-3.0 <= x && x < 3.0) return sinc(x) * sinc(x / 3);
    return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc; box edges are float32 like Pillow's.
// filter: MRCHIP_FILTER_BICUBIC (support 2) or MRCHIP_FILTER_LANCZOS (support 3)
int resample_coeffs(int filter, int in_size, float in0, float in1, int out_size, std::vector<int32_t> &bounds,
                    std::vector<int32_t> &kk) {
    double scale = (double)(in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = (filter == MRCHIP_FILTER_LANCZOS ? 3.0 : 2.0) * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; xx++) {
        double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; x++) {
            double wv = filter == MRCHIP_FILTER_LANCZOS ? lanczos_filter((x + xmin - center + 0.5) * ss)
                                                        : bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = wv; ww += wv;
        }
        for (x = 0; x < xmax; x++) if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        for (x = 0; x < ksize; x++) {
            double v = k[x] * (double)(1 << 22);
            kk[(size_t)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v) : (int32_t)(0.5 + v);
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

// ---- kernels ---------------------------------------------------------------------
__device__ __forceinline__ unsigned reduce_multiplier(int cells) {
    // Reduce.c division_UINT32(cells, 8): (UINT32)(2^32 / (256*cells)) evaluated in float32
    unsigned max_dividend = 256u * (unsigned)cells;
    float max_int = (float)(1 << 30) * 4.0f;
    return (unsigned)__fdiv_rn(max_int, (float)max_dividend);
}

__global__ __launch_bounds__(256) void reduce_kernel(const uint8_t *src, int spitch, size_t sstride, int w, int h, int c,
                                                     int fx, int fy, uint8_t *dst, int dpitch, size_t dstride, int ow,
                                                     int oh) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= ow) return;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const int y0 = oy * fy, y1 = min(h, y0 + fy), x0 = ox * fx, x1 = min(w, x0 + fx);
    const int cells = (y1 - y0) * (x1 - x0);
    const unsigned mult = reduce_multiplier(cells), amend = (unsigned)cells / 2;
    for (int ch = 0; ch < c; ch++) {
        unsigned ss = 0;
        for (int yy = y0; yy < y1; yy++)
            for (int xx = x0; xx < x1; xx++) ss += src[(size_t)yy * spitch + (size_t)xx * c + ch];
        dst[(size_t)oy * dpitch + (size_t)ox * c + ch] = (uint8_t)(((ss + amend) * mult) >> 24);
    }
}

// The same for the factors thumbnails actually use (2..4 across), four output pixels per lane: the FX*4*C input bytes
// of a row come in as FX*C aligned dwords (rows are 64-byte aligned and a lane's span is a multiple of 4 bytes), the
// 4*C output bytes leave as C dwords.  Lanes whose four pixels do not all have FX whole columns (right edge) and images
// with other factors take reduce_kernel's per-byte path.  The first version moved one byte per load: 0.6 TB/s, the
// dominant kernel of the 8000x6000 configuration.
template <int C, int FX>
__global__ __launch_bounds__(256) void reduce4_kernel(const uint8_t *src, int spitch, size_t sstride, int w, int h, int fy,
                                                      uint8_t *dst, int dpitch, size_t dstride, int ow, int oh) {
    const int g = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;       // group of 4 output pixels
    const int ox0 = 4 * g;
    if (ox0 >= ow) return;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const int y0 = oy * fy, y1 = min(h, y0 + fy);
    constexpr int NB = 4 * C, NDW = FX * C;
    if (ox0 + 4 <= ow && (ox0 + 4) * FX <= w) {
        unsigned acc[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) acc[j] = 0;
        for (int yy = y0; yy < y1; yy++) {
            const unsigned *rp = reinterpret_cast<const unsigned *>(src + (size_t)yy * spitch + (size_t)ox0 * FX * C);
            unsigned d[NDW];
#pragma unroll
            for (int q = 0; q < NDW; q++) d[q] = rp[q];
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int ch = 0; ch < C; ch++)
#pragma unroll
                    for (int i = 0; i < FX; i++) {
                        const int b = (p * FX + i) * C + ch;                 // byte of the lane's span (compile-time)
                        acc[p * C + ch] += (d[b >> 2] >> (8 * (b & 3))) & 0xffu;
                    }
        }
        const int cells = (y1 - y0) * FX;
        const unsigned mult = reduce_multiplier(cells), amend = (unsigned)cells / 2;
        unsigned o[C];
#pragma unroll
        for (int q = 0; q < C; q++) o[q] = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) o[j >> 2] |= (((acc[j] + amend) * mult) >> 24) << (8 * (j & 3));
        // (the reduced image is tight -- rows of rw*C bytes -- so its dwords may sit at any byte address)
        typedef unsigned __attribute__((aligned(1))) unsigned_unaligned;
        unsigned_unaligned *op = reinterpret_cast<unsigned_unaligned *>(dst + (size_t)oy * dpitch + (size_t)ox0 * C);
#pragma unroll
        for (int q = 0; q < C; q++) op[q] = o[q];
        return;
    }
    for (int ox = ox0; ox < min(ow, ox0 + 4); ox++) {                      // right edge: partial cells
        const int x0 = ox * FX, x1 = min(w, x0 + FX);
        const int cells = (y1 - y0) * (x1 - x0);
        const unsigned mult = reduce_multiplier(cells), amend = (unsigned)cells / 2;
        for (int ch = 0; ch < C; ch++) {
            unsigned ss = 0;
            for (int yy = y0; yy < y1; yy++)
                for (int xx = x0; xx < x1; xx++) ss += src[(size_t)yy * spitch + (size_t)xx * C + ch];
            dst[(size_t)oy * dpitch + (size_t)ox * C + ch] = (uint8_t)(((ss + amend) * mult) >> 24);
        }
    }
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= 22;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// unaligned 32-bit load (gfx950 global loads accept any byte address)
typedef unsigned __attribute__((aligned(1))) unsigned_u1;
__device__ __forceinline__ unsigned ld_u32(const uint8_t *p) { return *reinterpret_cast<const unsigned_u1 *>(p); }

// Horizontal pass: one lane per output column, RH consecutive rows per lane (the fixed-point
// coefficients of a column are loaded once and reused down the rows).  The coefficient table is
// stored transposed (kkT[tap][xx]) so that a wave's coefficient loads are contiguous; the window's
// bytes come in as (unaligned) dwords.  Taps beyond a window's length have zero coefficients, so
// the fixed MAXK-tap loop needs no masking (the bytes it over-reads lie inside the row pitch / PAD).
constexpr int RH = 8;
template <int C, int MAXK>
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t *src, int spitch, size_t sstride, int h,
                                                       uint8_t *dst, int dpitch, size_t dstride, int ow,
                                                       const int32_t *bounds, const int32_t *kkT) {
    const int xx = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * RH;
    if (xx >= ow) return;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const int xmin = bounds[2 * xx];
    int kv[MAXK];
#pragma unroll
    for (int x = 0; x < MAXK; x++) kv[x] = kkT[(size_t)x * ow + xx];
    constexpr int ND = (MAXK * C + 3) / 4;
    const int ny = min(RH, h - y0);
    for (int yy = 0; yy < ny; yy++) {
        const int y = y0 + yy;
        const uint8_t *row = src + (size_t)y * spitch + (size_t)xmin * C;
        unsigned v[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) v[d] = ld_u32(row + 4 * d);
        int ss[C];
#pragma unroll
        for (int ch = 0; ch < C; ch++) ss[ch] = 1 << 21;
#pragma unroll
        for (int x = 0; x < MAXK; x++) {
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                const int j = x * C + ch;
                ss[ch] += (int)((v[j >> 2] >> (8 * (j & 3))) & 0xffu) * kv[x];
            }
        }
        uint8_t *o = dst + (size_t)y * dpitch + (size_t)xx * C;
#pragma unroll
        for (int ch = 0; ch < C; ch++) o[ch] = clip8(ss[ch]);
    }
}

// LDS-staged variant: the 256 output columns of a workgroup read overlapping windows (stride
// scale*C bytes, window MAXK*C bytes), so the workgroup first copies the byte span it needs of each
// of its RH rows into LDS with coalesced aligned dword loads (all issued up front), then every lane
// assembles its window from aligned LDS dwords with v_alignbyte.  Used when the span fits LROW.
constexpr int RH_LROW = 4096;       // bytes of LDS per staged row
template <int C, int MAXK>
__global__ __launch_bounds__(256) void resize_h_lds_kernel(const uint8_t *src, int spitch, size_t sstride, int h,
                                                           uint8_t *dst, int dpitch, size_t dstride, int ow,
                                                           const int32_t *bounds, const int32_t *kkT) {
    __shared__ unsigned stage[RH][RH_LROW / 4];
    const int tid = threadIdx.x;
    const int xx0 = blockIdx.x * 256, y0 = blockIdx.y * RH;
    const int xx = xx0 + tid;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const int ny = min(RH, h - y0);
    const int xlast = min(xx0 + 255, ow - 1);
    const int first = bounds[2 * xx0];                                  // wave-uniform
    const int span_bytes = (bounds[2 * xlast] + MAXK - first) * C;      // bytes needed from `first`
    // ---- stage the rows ----
    unsigned delta[RH];                                                 // byte phase of each row's staging origin
#pragma unroll
    for (int r = 0; r < RH; r++) {
        const uint8_t *p = src + (size_t)min(y0 + r, h - 1) * spitch + (size_t)first * C;
        delta[r] = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3u);
    }
    const int ndw = (span_bytes + 3 + 3) / 4 + 1;
    for (int i = tid; i < ndw; i += 256) {
        unsigned v[RH];
#pragma unroll
        for (int r = 0; r < RH; r++) {
            const uint8_t *p = src + (size_t)min(y0 + r, h - 1) * spitch + (size_t)first * C - delta[r];
            v[r] = reinterpret_cast<const unsigned *>(p)[i];
        }
#pragma unroll
        for (int r = 0; r < RH; r++) stage[r][i] = v[r];
    }
    __syncthreads();
    if (xx >= ow) return;
    const int xmin = bounds[2 * xx];
    int kv[MAXK];
#pragma unroll
    for (int x = 0; x < MAXK; x++) kv[x] = kkT[(size_t)x * ow + xx];
    constexpr int ND = (MAXK * C + 3) / 4;
#pragma unroll
    for (int r = 0; r < RH; r++) {
        if (r < ny) {
            const unsigned off = (unsigned)(xmin - first) * C + delta[r];
            const unsigned *lp = &stage[r][off >> 2];
            const unsigned sh = off & 3u;
            unsigned raw[ND + 1], v[ND];
#pragma unroll
            for (int d = 0; d <= ND; d++) raw[d] = lp[d];
#pragma unroll
            for (int d = 0; d < ND; d++) v[d] = __builtin_amdgcn_alignbyte(raw[d + 1], raw[d], sh);
            int ss[C];
#pragma unroll
            for (int ch = 0; ch < C; ch++) ss[ch] = 1 << 21;
#pragma unroll
            for (int x = 0; x < MAXK; x++) {
#pragma unroll
                for (int ch = 0; ch < C; ch++) {
                    const int j = x * C + ch;
                    ss[ch] += (int)((v[j >> 2] >> (8 * (j & 3))) & 0xffu) * kv[x];
                }
            }
            uint8_t *o = dst + (size_t)(y0 + r) * dpitch + (size_t)xx * C;
#pragma unroll
            for (int ch = 0; ch < C; ch++) o[ch] = clip8(ss[ch]);
        }
    }
}

// generic fallback (any ksize), row-major table
template <int C>
__global__ __launch_bounds__(256) void resize_h_generic_kernel(const uint8_t *src, int spitch, size_t sstride, int h,
                                                               uint8_t *dst, int dpitch, size_t dstride, int ow,
                                                               const int32_t *bounds, const int32_t *kk, int ksize) {
    const int xx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xx >= ow) return;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int32_t *k = kk + (size_t)xx * ksize;
    const uint8_t *row = src + (size_t)y * spitch + (size_t)xmin * C;
    int ss[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) ss[ch] = 1 << 21;
    for (int x = 0; x < n; x++) {
        const int kv = k[x];
#pragma unroll
        for (int ch = 0; ch < C; ch++) ss[ch] += (int)row[x * C + ch] * kv;
    }
#pragma unroll
    for (int ch = 0; ch < C; ch++) dst[(size_t)y * dpitch + (size_t)xx * C + ch] = clip8(ss[ch]);
}

// Vertical pass: one lane per 4 adjacent bytes of an output row (unaligned dword loads; the
// per-row coefficients are wave-uniform -> scalar loads).
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t *src, int spitch, size_t sstride, int row_bytes,
                                                       uint8_t *dst, int dpitch, size_t dstride, int oh,
                                                       const int32_t *bounds, const int32_t *kk, int ksize) {
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4, yy = blockIdx.y;
    if (j >= row_bytes) return;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int32_t *k = kk + (size_t)yy * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21, s3 = 1 << 21;
    const uint8_t *p = src + (size_t)ymin * spitch + j;
    for (int y = 0; y < n; y++) {
        const unsigned v = ld_u32(p + (size_t)y * spitch);
        const int kv = k[y];
        s0 += (int)(v & 0xffu) * kv; s1 += (int)((v >> 8) & 0xffu) * kv;
        s2 += (int)((v >> 16) & 0xffu) * kv; s3 += (int)(v >> 24) * kv;
    }
    uint8_t *o = dst + (size_t)yy * dpitch + j;
    if (j + 4 <= row_bytes) {
        const unsigned r = (unsigned)clip8(s0) | ((unsigned)clip8(s1) << 8) | ((unsigned)clip8(s2) << 16) | ((unsigned)clip8(s3) << 24);
        *reinterpret_cast<unsigned_u1 *>(o) = r;
    } else {
        o[0] = clip8(s0);
        if (j + 1 < row_bytes) o[1] = clip8(s1);
        if (j + 2 < row_bytes) o[2] = clip8(s2);
    }
}

// ---- matrix-core resize pass ----------------------------------------------------------
// A bicubic pass is out[line][n] = clip8((2^21 + sum_k px[line][k] * coef[k][n]) >> 22): the
// product of the pixel lines (M x K, uint8) with a banded coefficient matrix (K x N, 22-bit
// fixed point), i.e. real multiply-accumulate work (13-17 taps per output byte) and the one stage
// of this path that belongs on the MFMA units.  v_mfma_i32_16x16x64_i8 multiplies signed bytes
// exactly into int32, so
//   * the pixels go in as px - 128 (one XOR 0x80 per dword); the constant 128 * sum(coef) + 2^21
//     comes back through `bias`,
//   * each coefficient is split into three balanced base-256 digits d0 + 256 d1 + 65536 d2
//     (|coef| < 2^23), one MFMA per digit, recombined with shifts (mod 2^32; the true sum fits).
// One wave owns a tile of 16 outputs (its B operand -- KB blocks of 64 input bytes x 3 digits --
// stays in registers) and walks down groups of 16 lines: per group KB 16-byte loads per lane
// (lane = line l&15, bytes (l>>4)*16.. of the block), 3*KB MFMAs, and the epilogue.  The MFMA
// result has lane = (output l&15, lines (l>>4)*4..+3), so the four result bytes of a lane are
// adjacent in the TRANSPOSED output: the horizontal pass writes its result transposed
// ([output byte][row]) as whole dwords, which makes the vertical pass the same kernel again
// (its K runs along the rows) and puts the final image back in row-major order.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef v4i v4i_u1 __attribute__((aligned(1)));

// Memory side of the pass.  The MFMA operand layout (lane = line, 16 consecutive K bytes) is the
// worst case for the vector memory path -- every lane of a load in a different row, 64 cache
// lines per instruction -- and so is the result layout for the stores.  Both go through a small
// wave-private LDS panel instead (no workgroup barrier: a wave's LDS operations execute in order):
//   in : lanes run along K first (KB*4 chunks of 16 bytes per line), i.e. each load instruction
//        covers 64/(KB*4) lines x KB*64 contiguous bytes; the panel rows are 16-byte aligned
//        whatever kbase is, so the operand reads are aligned ds_read_b128
//   out: the dwords of four line groups are collected per output and leave as 16-byte stores,
//        64 contiguous bytes per output and quad.
template <int KB, int WPB>
__global__ __launch_bounds__(64 * WPB) void resize_mm_kernel(const uint8_t *src, int spitch, size_t sstride, int nlines,
                                                        uint8_t *dst, int dpitch, size_t dstride, int nout, int ntiles,
                                                        const int32_t *kbase, const int32_t *bias, const v4i *btab,
                                                        int quads_per_wave, int pad_ok, int gx, int gy, int gz) {
    // XCD-aware work order: workgroup L runs on XCD L % 8 (round-robin dispatch).  Neighbouring tiles
    // read overlapping bytes of the same lines, so each XCD (own L2) takes one contiguous range of the
    // x-fastest work list instead of every eighth item.
    const int total = gx * gy * gz, per = (total + 7) >> 3;
    const int V = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (V >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int bx = V % gx, by = (V / gx) % gy, bz = V / (gx * gy);
    constexpr int CPL = KB * 4;                   // 16-byte chunks per line
    constexpr int LPI = 64 / CPL;                 // lines per load instruction
    constexpr int LSTR = KB * 64 + 16;            // panel row stride: 8 consecutive rows hit 8 distinct 16-byte bank groups
    constexpr int OSTR = 80;
    __shared__ __attribute__((aligned(16))) unsigned char inP[WPB][16 * LSTR];
    __shared__ __attribute__((aligned(16))) unsigned char outP[WPB][16 * OSTR];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = bx * WPB + wv;
    if (tile >= ntiles) return;
    src += (size_t)bz * sstride;
    dst += (size_t)bz * dstride;
    const int nn = lane & 15, kq = lane >> 4;
    v4i B[KB][3];
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int d = 0; d < 3; d++) B[kb][d] = btab[((size_t)(tile * KB + kb) * 3 + d) * 64 + lane];
    const int bs = bias[tile * 16 + nn];                  // padded to 16 * ntiles entries
    const int kb0 = kbase[tile];
    const int nquads = (nlines + 63) >> 6;
    const int q0 = by * quads_per_wave, q1 = min(nquads, q0 + quads_per_wave);
    if (q0 >= q1) return;
    unsigned char *ip = inP[wv], *op = outP[wv];
    const int ld_line = lane / CPL, ld_chunk = lane % CPL;

    auto gload = [&](int g, v4i (&A)[KB]) {
#pragma unroll
        for (int r = 0; r < KB; r++) {
            const int line = min(g * 16 + ld_line + r * LPI, nlines - 1);   // past the end: repeat the last line (never stored)
            A[r] = *reinterpret_cast<const v4i_u1 *>(src + (size_t)line * spitch + kb0 + ld_chunk * 16);
        }
    };
    v4i stage[2][KB];                                     // two groups of lines in flight
    const int g_end = q1 * 4;
    gload(q0 * 4, stage[0]);
    gload(min(q0 * 4 + 1, g_end - 1), stage[1]);
    for (int q = q0; q < q1; q++) {
#pragma unroll
    for (int gi = 0; gi < 4; gi++) {
        const int g = q * 4 + gi;
        // this group's lines: registers -> panel (px - 128 on the way)
#pragma unroll
        for (int r = 0; r < KB; r++)
            *reinterpret_cast<v4i *>(ip + (ld_line + r * LPI) * LSTR + ld_chunk * 16) = stage[gi & 1][r] ^ (int)0x80808080;
        gload(min(g + 2, g_end - 1), stage[gi & 1]);      // lines of group g+2 are in flight during the MFMAs
        lds_wave_sync();
        v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            const v4i x = *reinterpret_cast<const v4i *>(ip + nn * LSTR + kb * 64 + kq * 16);
            acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][2], acc2, 0, 0, 0);
        }
        unsigned packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ss = (int)((unsigned)acc0[i] + ((unsigned)acc1[i] << 8) + ((unsigned)acc2[i] << 16) + (unsigned)bs);
            int q = min(max(ss >> 22, 0), 255);                              // Resample.c clip8
            // keep the compiler from fusing two of these into v_ashr_pk_u8_i32: hipcc 7.2 assumes that
            // instruction clears the upper half of its destination, gfx950 leaves the old bits there
            asm volatile("" : "+v"(q));
            packed |= (unsigned)q << (8 * i);
        }
        *reinterpret_cast<unsigned *>(op + nn * OSTR + gi * 16 + kq * 4) = packed;
        if (gi == 3) {
            lds_wave_sync();
            const int n2 = tile * 16 + (lane >> 2), c16 = (lane & 3) * 16;
            const v4i v = *reinterpret_cast<const v4i *>(op + (lane >> 2) * OSTR + c16);
            const int line0 = q * 64 + c16;
            if (n2 < nout) {
                uint8_t *o = dst + (size_t)n2 * dpitch + line0;
                if (pad_ok || line0 + 16 <= nlines) *reinterpret_cast<v4i_u1 *>(o) = v;
                else
                    for (int i = 0; line0 + i < nlines; i++) o[i] = (uint8_t)((unsigned)v[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
    }
}

// Workgroup-wide variant of the memory side: the 16 tiles of a workgroup read overlapping 128-byte
// windows of the same 16 lines (tiles are 48 input bytes apart at scale 3), so the workgroup loads
// the union once -- one contiguous span per line, lanes along it -- into a double-buffered LDS
// panel (one barrier per line group) and every wave takes its operands from there.  2.4x fewer
// load instructions, no overlap re-fetched.  Needs 16-byte aligned tile bases (build_mm) and a
// span of at most 1024 bytes.
constexpr int PNW = 16;         // waves = tiles per workgroup (widest form; 8 where the narrower panel fits its 512 loader lanes)
template <int KB, int NW>
__global__ __launch_bounds__(64 * NW) void resize_mm_panel_kernel(const uint8_t *src, int spitch, size_t sstride, int nlines,
                                                                   uint8_t *dst, int dpitch, size_t dstride, int nout, int ntiles,
                                                                   const int32_t *kbase, const int32_t *bias, const v4i *btab,
                                                                   int quads_per_wave, int pad_ok, int gx, int gy, int gz,
                                                                   int panel_w, int pws, int line_bytes) {
    const int total = gx * gy * gz, per = (total + 7) >> 3;            // XCD-contiguous work order (see above)
    const int V = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (V >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int bx = V % gx, by = (V / gx) % gy, bz = V / (gx * gy);
    constexpr int OSTR = 80;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *inP = smem;                                        // [2][16][pws]
    unsigned char *outP = smem + 2 * 16 * pws;                        // [NW][16][OSTR]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile0 = bx * NW, tile = tile0 + wv;
    const bool active = tile < ntiles;
    src += (size_t)bz * sstride;
    dst += (size_t)bz * dstride;
    const int nn = lane & 15, kq = lane >> 4;
    v4i B[KB][3];
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int d = 0; d < 3; d++) B[kb][d] = active ? btab[((size_t)(tile * KB + kb) * 3 + d) * 64 + lane] : (v4i){0, 0, 0, 0};
    const int bs = active ? bias[tile * 16 + nn] : 0;
    const int kbP = kbase[tile0];                                     // first byte of the panel
    const int koff = active ? kbase[tile] - kbP : 0;                  // multiple of 16
    const int nquads = (nlines + 63) >> 6;
    const int q0 = by * quads_per_wave, q1 = min(nquads, q0 + quads_per_wave);
    if (q0 >= q1) return;                                             // uniform over the workgroup
    unsigned char *op = outP + wv * 16 * OSTR;
    const int cpl = panel_w >> 4;                                     // 16-byte chunks per line
    const int ld_line = tid / cpl, ld_chunk = tid - ld_line * cpl;
    // chunks that start past the end of a line hold no tap (their coefficients are zero) and are not loaded:
    // the panel is as wide as the widest workgroup needs, the last workgroups of a line would otherwise read
    // up to panel_w bytes past the line -- past the allocation for the last line of a small buffer
    const bool loader = tid < 16 * cpl && kbP + ld_chunk * 16 < line_bytes;

    auto gload = [&](int g) -> v4i {
        const int line = min(g * 16 + ld_line, nlines - 1);           // past the end: repeat the last line (never stored)
        return *reinterpret_cast<const v4i_u1 *>(src + (size_t)line * spitch + kbP + ld_chunk * 16);
    };
    const int g_end = q1 * 4;
    v4i stage[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (loader) { stage[0] = gload(q0 * 4); stage[1] = gload(min(q0 * 4 + 1, g_end - 1)); }
    for (int q = q0; q < q1; q++) {
#pragma unroll
    for (int gi = 0; gi < 4; gi++) {
        const int g = q * 4 + gi;
        unsigned char *ip = inP + (gi & 1) * 16 * pws;
        if (loader) {
            *reinterpret_cast<v4i *>(ip + ld_line * pws + ld_chunk * 16) = stage[gi & 1] ^ (int)0x80808080;   // px - 128
            stage[gi & 1] = gload(min(g + 2, g_end - 1));             // group g+2 in flight during the MFMAs
        }
        // one barrier per group: buffer gi&1 is rewritten two groups later, after the barrier of group
        // g+1, which every wave reaches only when it is done reading this one
        lds_barrier();
        if (active) {
            v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const v4i x = *reinterpret_cast<const v4i *>(ip + nn * pws + koff + kb * 64 + kq * 16);
                acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][2], acc2, 0, 0, 0);
            }
            unsigned packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ss = (int)((unsigned)acc0[i] + ((unsigned)acc1[i] << 8) + ((unsigned)acc2[i] << 16) + (unsigned)bs);
                int qv = min(max(ss >> 22, 0), 255);                         // Resample.c clip8
                asm volatile("" : "+v"(qv));                                 // no v_ashr_pk_u8_i32 (see resize_mm_kernel)
                packed |= (unsigned)qv << (8 * i);
            }
            *reinterpret_cast<unsigned *>(op + nn * OSTR + gi * 16 + kq * 4) = packed;
            if (gi == 3) {
                lds_wave_sync();
                const int n2 = tile * 16 + (lane >> 2), c16 = (lane & 3) * 16;
                const v4i v = *reinterpret_cast<const v4i *>(op + (lane >> 2) * OSTR + c16);
                const int line0 = q * 64 + c16;
                if (n2 < nout) {
                    uint8_t *o = dst + (size_t)n2 * dpitch + line0;
                    if (pad_ok || line0 + 16 <= nlines) *reinterpret_cast<v4i_u1 *>(o) = v;
                    else
                        for (int i = 0; line0 + i < nlines; i++) o[i] = (uint8_t)((unsigned)v[i >> 2] >> (8 * (i & 3)));
                }
            }
        }
    }
    }
}

// Four results of a pass -> four bytes: clip8(s >> 22) of each, packed.  v_ashr_pk_u8_i32 does a pair per instruction
// ({sat_u8(S0 >> n), sat_u8(S1 >> n)} into one half of the destination, the other half keeps its bits -- which is why
// the compiler must not form it on its own, see resize_mm_kernel: hipcc 7.2 takes the other half for zero); op_sel[3]
// picks the upper half.  Measured on the part (byte order, saturation at both ends, the preserved half).
__device__ __forceinline__ unsigned clip8x4_shr22(const int (&s)[4]) {
    unsigned d = 0;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "+v"(d) : "v"(s[0]), "v"(s[1]));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 22 op_sel:[0,0,0,1]" : "+v"(d) : "v"(s[2]), "v"(s[3]));
    return d;
}

// Both passes in one workgroup: the horizontal pass of a run of lines stays in LDS (transposed, one
// row per output byte, px - 128 like the panel) and the vertical pass reads its operands from there,
// so the pass-to-pass image (a third of the page at scale 3, written and read back by the two-kernel
// form) never goes to memory.  A workgroup = 8 horizontal tiles (128 output bytes) x NB blocks of RV
// vertical tiles (16 RV output rows each).  It streams the lines those rows have taps on in groups of
// 16 exactly as the panel kernel does; the transposed rows are a ring of 17 chunks of 16 lines (a
// block spans at most 256 lines, ThumbPlan_build), so consecutive blocks share the lines of the
// filter support instead of reading them again.  When the last line of a block is in, wave w takes
// vertical tile w % RV and every (8 / RV)th horizontal tile: one vertical B operand per wave, used
// 8 / RV times.  Chunks of a 64-line operand block that lie past the block's last line hold older
// lines; their coefficients are zero.
template <int KBH, int KBV, int RV>
__global__ __launch_bounds__(512) void resize_mm_fused_kernel(const uint8_t *src, int spitch, size_t sstride, int nlines,
                                                               int line_bytes, uint8_t *dst, int dpitch, size_t dstride,
                                                               int nout_h, int ntiles_h, const int32_t *kbase_h,
                                                               const int32_t *bias_h, const v4i *bt_h, int nout_v,
                                                               int ntiles_v, const int32_t *kbase_v, const int32_t *kend_v,
                                                               const int32_t *bias_v, const v4i *bt_v, int gx, int gy, int gz,
                                                               int panel_w, int pws, int nb, const uint8_t *alt, int apitch,
                                                               size_t astride, const unsigned *rowmap, int rmwords) {
    // alt / rowmap (optional): line y of page z comes from `src` where bit y of rowmap[z * rmwords ..] is set and from
    // `alt` (same layout, its own pitch) where it is clear -- the bg layer of optimise's band walkers holds only the rows
    // of its bands, every other row IS the image row (k_optimise.hip, OptBand): they are read from the image in place
    constexpr int NW = 8, TSTR = 272, RING = TSTR / 16;
    const int total = gx * gy * gz, per = (total + 7) >> 3;            // XCD-contiguous work order (see resize_mm_kernel)
    const int V = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (V >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int bx = V % gx, by = (V / gx) % gy, bz = V / (gx * gy);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *inP = smem;                                        // [2][16][pws]
    unsigned char *T = smem + 2 * 16 * pws;                           // [128][TSTR]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nn = lane & 15, kq = lane >> 4;
    src += (size_t)bz * sstride;
    dst += (size_t)bz * dstride;
    unsigned *rmap = reinterpret_cast<unsigned *>(smem + 2 * 16 * pws + 128 * TSTR);      // the page's row map (rowmap != nullptr)
    // (without a row map every bit is set and `alt` is `src`: ONE load site either way -- a branch around two would cost
    // the compiler its count of the loads in flight, i.e. a full wait per group)
    if (rowmap) alt += (size_t)bz * astride; else { alt = src; apitch = spitch; }
    for (int i = tid; i < rmwords; i += 512) rmap[i] = rowmap ? rowmap[(size_t)bz * rmwords + i] : 0xffffffffu;
    __syncthreads();
    const int tile0 = bx * NW;
    const int tvA = by * nb * RV, tvZ = min(tvA + nb * RV, ntiles_v);   // vertical tiles of this workgroup
    const int Lb = kbase_v[tvA];                                      // multiple of 16
    const int gtot = max((min(kend_v[tvZ - 1], nlines) - Lb + 15) >> 4, 1);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);               // wave-uniform copy (scalar register)
    const int tile = tile0 + wvu;
    const bool active = tile < ntiles_h;
    // the horizontal B operand is loaded again after every vertical block: its registers carry the vertical operand
    // meanwhile, which keeps the kernel at six waves per SIMD (three workgroups per CU)
    v4i B[KBH][3];
    auto load_bh = [&]() {
#pragma unroll
        for (int kb = 0; kb < KBH; kb++)
#pragma unroll
            for (int d = 0; d < 3; d++)
                B[kb][d] = bt_h[((size_t)(min(tile, ntiles_h - 1) * KBH + kb) * 3 + d) * 64 + lane];
        // waited for here, once per block, not at its first use inside the line loop (where the wait would take the
        // loads of the coming groups with it)
#pragma unroll
        for (int kb = 0; kb < KBH; kb++)
#pragma unroll
            for (int d = 0; d < 3; d++) asm volatile("" : "+v"(B[kb][d]));
    };
    load_bh();
    const int bs = active ? bias_h[tile * 16 + nn] : 0;
    const int kbP = kbase_h[tile0];
    const int koff = active ? kbase_h[tile] - kbP : 0;
    // Every lane loads and stores in every group, with no branch around either: the compiler then counts the loads
    // in flight (s_waitcnt vmcnt(1) for the group at hand while the next one is on its way) instead of waiting for
    // all of them.  Lanes beyond the 16 x cpl chunks of a group repeat the first ones (same bytes to the same place);
    // chunks that start past the end of a line hold no tap (zero coefficients) and take the line's last chunk instead,
    // which keeps the reads inside the line (see resize_mm_panel_kernel).
    const int cpl = panel_w >> 4;
    const int lt = tid % (16 * cpl);
    const int ld_line = lt / cpl, ld_chunk = lt - ld_line * cpl;
    const unsigned colofs = kbP + min(ld_chunk, (line_bytes - 1 - kbP) >> 4) * 16;
    const int pofs = ld_line * pws + ld_chunk * 16;
    auto gload = [&](int g) -> v4i {                                  // 32-bit lane offset from the (scalar) page base
        const int line = min(Lb + g * 16 + ld_line, nlines - 1);
        const bool own = (rmap[line >> 5] >> (line & 31)) & 1u;
        const uint8_t *base = own ? src : alt;
        const unsigned pit = own ? (unsigned)spitch : (unsigned)apitch;
        return *reinterpret_cast<const v4i_u1 *>(base + ((unsigned)line * pit + colofs));
    };
    v4i stage[2];
    stage[0] = gload(0);
    stage[1] = gload(min(1, gtot - 1));
    unsigned char *trow = T + (wv * 16 + nn) * TSTR + kq * 4;

    auto group = [&](int g, auto par) {
        constexpr int P = decltype(par)::value;
        unsigned char *ip = inP + P * 16 * pws;
        *reinterpret_cast<v4i *>(ip + pofs) = stage[P] ^ (int)0x80808080;
        stage[P] = gload(min(g + 2, gtot - 1));
        lds_barrier();
        if (active) {
            v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < KBH; kb++) {
                const v4i x = *reinterpret_cast<const v4i *>(ip + nn * pws + koff + kb * 64 + kq * 16);
                acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, B[kb][2], acc2, 0, 0, 0);
            }
            int ss[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                ss[i] = (int)((unsigned)acc0[i] + ((unsigned)acc1[i] << 8) + ((unsigned)acc2[i] << 16) + (unsigned)bs);
            const unsigned packed = clip8x4_shr22(ss);                   // Resample.c clip8 (the 8-bit image between the passes)
            // lines 16 g + 4 kq .. + 3 of output byte nn: row stride 68 dwords -> the 64 lanes hit 64 banks
            *reinterpret_cast<unsigned *>(trow + (g % RING) * 16) = packed ^ 0x80808080u;
        }
    };
    // vertical pass of the block of tiles tv0 .. tv0 + RV - 1 (all their lines are in the ring)
    // vertical pass of the block of tiles tv0 .. tv0 + RV - 1 (all their lines are in the ring)
    auto vblock = [&](int tv0) {
        lds_barrier();
        // lane indices behind an opaque copy: nothing of this block is computed ahead and kept in registers across the line loop
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        const int nv = lane_v & 15, kv = lane_v >> 4;
        const int tv = tv0 + wvu % RV;
        if (tv >= tvZ) return;
        v4i Bv[KBV][3];
#pragma unroll
        for (int kb = 0; kb < KBV; kb++)
#pragma unroll
            for (int d = 0; d < 3; d++) Bv[kb][d] = bt_v[((size_t)(tv * KBV + kb) * 3 + d) * 64 + lane_v];
        const int bsv = bias_v[tv * 16 + nv];
        const int c0 = ((kbase_v[tv] - Lb) >> 4) + kv;                // first chunk of this lane, before the ring wraps
        int coff[KBV];
#pragma unroll
        for (int kb = 0; kb < KBV; kb++) coff[kb] = ((c0 + kb * 4) % RING) * 16;
        const int row = tv * 16 + nv;
#pragma unroll 1
        for (int ht = wvu / RV; ht < NW; ht += NW / RV) {
            if (tile0 + ht >= ntiles_h) break;
            v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < KBV; kb++) {
                const v4i x = *reinterpret_cast<const v4i *>(T + (ht * 16 + nv) * TSTR + coff[kb]);
                acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, Bv[kb][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, Bv[kb][1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, Bv[kb][2], acc2, 0, 0, 0);
            }
            int ss[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                ss[i] = (int)((unsigned)acc0[i] + ((unsigned)acc1[i] << 8) + ((unsigned)acc2[i] << 16) + (unsigned)bsv);
            const unsigned packed = clip8x4_shr22(ss);
            // output row 16 tv + nn, bytes 16 (tile0 + ht) + 4 kq .. + 3 (the eight waves fill 128-byte runs of a row)
            const int b0 = (tile0 + ht) * 16 + kv * 4;
            if (row < nout_v && b0 < nout_h) {
                uint8_t *o = dst + ((unsigned)row * (unsigned)dpitch + (unsigned)b0);
                typedef unsigned u32_u1 __attribute__((aligned(1)));
                if (b0 + 4 <= nout_h) *reinterpret_cast<u32_u1 *>(o) = packed;
                else
                    for (int i = 0; b0 + i < nout_h; i++) o[i] = (uint8_t)(packed >> (8 * i));
            }
        }
    };
    // the line stream may be one group past a block's last line when its vertical pass runs (groups go in
    // pairs): the ring has room, a block spans at most 16 of its 17 chunks
    int tvb = tvA;                                                     // next block to finish
    auto gend = [&](int tv0) { return max((min(kend_v[min(tv0 + RV, tvZ) - 1], nlines) - Lb + 15) >> 4, 1); };
    int ge = gend(tvb);
    // (always whole pairs, so that the loads in flight are the same two at the top of every iteration: a last odd
    // group is followed by a repeat of itself into the ring's spare chunk)
    for (int g = 0; g < gtot; g += 2) {
        group(g, std::integral_constant<int, 0>());
        group(g + 1, std::integral_constant<int, 1>());
        while (tvb < tvZ && g + 2 >= ge) {
            vblock(tvb);
            tvb += RV;
            ge = tvb < tvZ ? gend(tvb) : INT_MAX;
            load_bh();
        }
    }
}

// B operand, kbase and bias of one pass.  cs: byte stride between the taps of one output (the
// channel count for the horizontal pass over interleaved pixels, 1 for the vertical pass).
static bool build_mm(const std::vector<int32_t> &bounds, const std::vector<int32_t> &kk, int ksize, int nout_px, int cs,
                     ThumbPlan::Mm &t) {
    t = ThumbPlan::Mm();
    t.nout = nout_px * cs;
    t.ntiles = cdiv(t.nout, 16);
    t.kbase.assign(t.ntiles, 0);
    t.kend.assign(t.ntiles, 0);
    t.bias.assign((size_t)t.ntiles * 16, 0);
    // first input byte of a tile, aligned down to 16 bytes when two 64-byte blocks still cover the tile
    // (aligned bases allow the workgroup-wide panel loads; a third MFMA block is not worth it), else as
    // far as it costs no extra block
    int KB = 1;
    for (int kalign = 16; kalign >= 1; kalign /= 4) {          // 16, 4, 1
        KB = 1;
        for (int T = 0; T < t.ntiles; T++) {
            int lo = INT_MAX, hi = -1;
            for (int j = 16 * T; j < std::min(16 * T + 16, t.nout); j++) {
                const int xo = j / cs, ch = j % cs, xmin = bounds[2 * xo], xs = bounds[2 * xo + 1];
                if (xs <= 0) continue;
                lo = std::min(lo, xmin * cs + ch);
                hi = std::max(hi, (xmin + xs - 1) * cs + ch);
            }
            if (hi < 0) { lo = 0; hi = 0; }
            lo &= ~(kalign - 1);
            t.kbase[T] = lo;
            t.kend[T] = hi + 1;
            KB = std::max(KB, cdiv(hi - lo + 1, 64));
        }
        int KB1 = 1;                                            // blocks needed without any alignment
        for (int T = 0; T < t.ntiles; T++) {
            int lo = INT_MAX, hi = -1;
            for (int j = 16 * T; j < std::min(16 * T + 16, t.nout); j++) {
                const int xo = j / cs, ch = j % cs, xmin = bounds[2 * xo], xs = bounds[2 * xo + 1];
                if (xs <= 0) continue;
                lo = std::min(lo, xmin * cs + ch);
                hi = std::max(hi, (xmin + xs - 1) * cs + ch);
            }
            if (hi >= 0) KB1 = std::max(KB1, cdiv(hi - lo + 1, 64));
        }
        t.kalign = kalign;
        if (KB == KB1 || (kalign == 16 && KB <= 2)) break;     // a second block is cheaper than losing the panel loads
    }
    if (KB > 2) return false;
    t.KB = KB;
    for (int T = 1; T < t.ntiles; T++) t.kend[T] = std::max(t.kend[T], t.kend[T - 1]);
    // span of the 16 tiles of a workgroup (panel kernel)
    t.panel_w = 0;
    t.panel_w8 = 0;
    for (int T0 = 0; T0 < t.ntiles; T0 += 16) {
        const int T1 = std::min(T0 + 15, t.ntiles - 1);
        t.panel_w = std::max(t.panel_w, t.kbase[T1] + KB * 64 - t.kbase[T0]);
    }
    for (int T0 = 0; T0 < t.ntiles; T0 += 8) {
        const int T1 = std::min(T0 + 7, t.ntiles - 1);
        t.panel_w8 = std::max(t.panel_w8, t.kbase[T1] + KB * 64 - t.kbase[T0]);
    }
    t.panel_w = round_up(t.panel_w, 16);
    t.panel_w8 = round_up(t.panel_w8, 16);
    t.b.assign((size_t)t.ntiles * KB * 3 * 1024, 0);
    for (int T = 0; T < t.ntiles; T++)
        for (int nn = 0; nn < 16; nn++) {
            const int j = 16 * T + nn;
            if (j >= t.nout) continue;
            const int xo = j / cs, ch = j % cs, xmin = bounds[2 * xo], xs = bounds[2 * xo + 1];
            long long sum = 0;
            for (int x = 0; x < xs; x++) {
                const int coef = kk[(size_t)xo * ksize + x];
                sum += coef;
                const int k = (xmin + x) * cs + ch - t.kbase[T];
                const int kb = k / 64, kq = (k % 64) / 16, bb = k % 16, lane = kq * 16 + nn;
                const int d0 = ((coef + 128) & 255) - 128, c1 = (coef - d0) >> 8;
                const int d1 = ((c1 + 128) & 255) - 128, d2 = (c1 - d1) >> 8;
                if (d2 < -128 || d2 > 127) return false;
                const int dg[3] = {d0, d1, d2};
                for (int d = 0; d < 3; d++)
                    t.b[(((size_t)(T * KB + kb) * 3 + d) * 64 + lane) * 16 + bb] = (unsigned char)(signed char)dg[d];
            }
            const long long bv = (1ll << 21) + 128 * sum;
            if (bv > INT_MAX || bv < INT_MIN) return false;
            t.bias[j] = (int32_t)bv;
        }
    return true;
}

// Plan of one thumbnail: host tables + scratch sizes.  Built once per (shape, request).
int ThumbPlan_build(ThumbPlan &p, int w, int h, int c, int req_w, int req_h, int filter, double reducing_gap) {
    p = ThumbPlan();
    p.w = w; p.h = h; p.c = c;
    p.filter = filter; p.reducing_gap = reducing_gap;
    if (filter != MRCHIP_FILTER_BICUBIC && filter != MRCHIP_FILTER_LANCZOS) {
        set_error("thumbnail: unknown filter %d", filter);
        return MRCHIP_E_ARG;
    }
    p.changed = thumbnail_size(w, h, req_w, req_h, &p.ow, &p.oh);
    if (!p.changed) return 0;
    // Image.resize: factor = int(extent / size / reducing_gap) or 1; reducing_gap None (<= 0 here): no reduce
    p.fx = p.fy = 1;
    if (reducing_gap > 0) {
        p.fx = (int)((double)w / p.ow / reducing_gap); if (p.fx < 1) p.fx = 1;
        p.fy = (int)((double)h / p.oh / reducing_gap); if (p.fy < 1) p.fy = 1;
    }
    float bw = (float)w, bh = (float)h;
    p.rw = w; p.rh = h;
    if (p.fx > 1 || p.fy > 1) {
        p.rw = (w + p.fx - 1) / p.fx; p.rh = (h + p.fy - 1) / p.fy;
        bw = (float)((double)w / p.fx); bh = (float)((double)h / p.fy);
    }
    p.need_h = (p.ow != p.rw) || (bw != (float)p.ow);
    p.need_v = (p.oh != p.rh) || (bh != (float)p.oh);
    if (p.need_h) {
        p.ksh = resample_coeffs(filter, p.rw, 0.f, bw, p.ow, p.bh_, p.kh_);
        if (p.ksh <= THUMB_MAXK) {                 // transposed, zero-padded to THUMB_MAXK taps
            p.khT_.assign((size_t)THUMB_MAXK * p.ow, 0);
            for (int xx = 0; xx < p.ow; xx++)
                for (int x = 0; x < p.ksh; x++) p.khT_[(size_t)x * p.ow + xx] = p.kh_[(size_t)xx * p.ksh + x];
        }
    }
    if (p.need_v) p.ksv = resample_coeffs(filter, p.rh, 0.f, bh, p.oh, p.bv_, p.kv_);
    p.mm_ok = p.need_h && p.need_v && !getenv("MRCHIP_THUMB_NO_MFMA") &&
              build_mm(p.bh_, p.kh_, p.ksh, p.ow, c, p.mmh) && build_mm(p.bv_, p.kv_, p.ksv, p.oh, 1, p.mmv);
    // one blob: [bh, kh, bv, kv, khT | per pass: kbase, bias, B operand (16-byte aligned)]
    auto put = [&](const void *d, size_t bytes) {
        size_t off = (p.blob_.size() + 15) & ~(size_t)15;
        p.blob_.resize(off + bytes);
        if (bytes) memcpy(p.blob_.data() + off, d, bytes);
        return off;
    };
    p.off_bh = put(p.bh_.data(), p.bh_.size() * 4);
    p.off_kh = put(p.kh_.data(), p.kh_.size() * 4);
    p.off_bv = put(p.bv_.data(), p.bv_.size() * 4);
    p.off_kv = put(p.kv_.data(), p.kv_.size() * 4);
    p.off_khT = put(p.khT_.data(), p.khT_.size() * 4);
    if (p.mm_ok) {
        ThumbPlan::Mm *mm[2] = {&p.mmh, &p.mmv};
        for (int i = 0; i < 2; i++) {
            p.off_mm[i][0] = put(mm[i]->kbase.data(), mm[i]->kbase.size() * 4);
            p.off_mm[i][1] = put(mm[i]->bias.data(), mm[i]->bias.size() * 4);
            p.off_mm[i][2] = put(mm[i]->b.data(), mm[i]->b.size());
            p.off_mmend[i] = put(mm[i]->kend.data(), mm[i]->kend.size() * 4);
        }
        // both passes in one kernel (resize_mm_fused_kernel): the horizontal panel must fit its 512 loader lanes
        // and the vertical tile bases must be 16-byte aligned (they index LDS rows read as 16-byte operands);
        // RV = vertical tiles per block, as many as keep the lines of a block within 16 of the 17 chunks of the LDS ring
        p.fuse_rv = 0;
        if (!getenv("MRCHIP_THUMB_NO_FUSE") && p.mmh.kalign == 16 && p.mmh.panel_w8 <= 512 && p.mmv.kalign == 16) {
            for (int rv = 4; rv >= 1 && !p.fuse_rv; rv /= 2) {
                int span = 0;
                for (int t0 = 0; t0 < p.mmv.ntiles; t0 += rv) {
                    const int t1 = std::min(t0 + rv, p.mmv.ntiles) - 1;
                    span = std::max(span, round_up(p.mmv.kend[t1] - p.mmv.kbase[t0], 16));
                }
                if (span <= 256) p.fuse_rv = rv;
            }
        }
    }
    p.blob_.resize(p.blob_.size() + 64);
    return 0;
}

size_t ThumbPlan_table_bytes(const ThumbPlan &p) { return p.blob_.size(); }

void ThumbPlan_scratch2_dims(const ThumbPlan &p, int *width_bytes, int *rows) {
    if (p.mm_ok) {           // transposed: one line per output byte of the horizontal pass, K slack at the end
        *width_bytes = round_up(p.rh, 64) + 128;
        *rows = p.mmh.ntiles * 16;
    } else {
        *width_bytes = p.ow * p.c;
        *rows = p.rh;
    }
}

// d_tables: device copy of [bh_, kh_, bv_, kv_] in that order (int32), made by the caller.
// scratch1: rw*rh*c per page (reduce output, if any); scratch2: ow*rh*c per page (horizontal pass output)
// true when launch_thumbnail_plan will read `src` with the one-kernel matrix-core form (which can take rows from a second
// plane: ThumbAlt)
bool thumbnail_reads_source_fused(const ThumbPlan &p, Plane src, Plane dst) {
    const bool red = p.fx > 1 || p.fy > 1;
    return p.changed && !red && p.mm_ok && p.fuse_rv && (size_t)p.h * src.pitch < ((size_t)1 << 32) &&
           (size_t)p.oh * dst.pitch < ((size_t)1 << 32);
}

int launch_thumbnail_plan(mrchip_ctx *ctx, hipStream_t s, const ThumbPlan &p, Plane src, Plane dst,
                          const void *d_tables, Plane scratch1, Plane scratch2, int npages, const ThumbAlt *ta) {
    const int c = p.c;
    if (ta && !thumbnail_reads_source_fused(p, src, dst)) {
        set_error("thumbnail: a row map needs the one-kernel form (thumbnail_reads_source_fused)");
        return MRCHIP_E_ARG;
    }
    if (!p.changed) {
        for (int i = 0; i < npages; i++)
            HIP_TRY(hipMemcpy2DAsync(dst.page(i), dst.pitch, src.page(i), src.pitch, (size_t)p.w * c, p.h,
                                     hipMemcpyDeviceToDevice, s));
        return 0;
    }
    const double alg = ((double)c * p.w * p.h + (double)c * p.ow * p.oh) * npages;
    Plane cur = src;
    int cw = p.w, ch_ = p.h;
    const bool red = p.fx > 1 || p.fy > 1;
    if (red) {
        const dim3 g4(cdiv(cdiv(p.rw, 4), 256), p.rh, npages);
#define RED4(CC, FXX)                                                                                                   \
    LAUNCH(ctx, s, "thumb_reduce", alg,                                                                                 \
           hipLaunchKernelGGL((reduce4_kernel<CC, FXX>), g4, dim3(256), 0, s, cur.p, cur.pitch, cur.stride, cw, ch_, p.fy, \
                              scratch1.p, scratch1.pitch, scratch1.stride, p.rw, p.rh))
        if (c == 3 && p.fx == 2) RED4(3, 2);
        else if (c == 3 && p.fx == 3) RED4(3, 3);
        else if (c == 3 && p.fx == 4) RED4(3, 4);
        else if (c == 1 && p.fx == 2) RED4(1, 2);
        else if (c == 1 && p.fx == 3) RED4(1, 3);
        else if (c == 1 && p.fx == 4) RED4(1, 4);
        else
            LAUNCH(ctx, s, "thumb_reduce", alg,
                   hipLaunchKernelGGL(reduce_kernel, dim3(cdiv(p.rw, 256), p.rh, npages), dim3(256), 0, s, cur.p, cur.pitch,
                                      cur.stride, cw, ch_, c, p.fx, p.fy, scratch1.p, scratch1.pitch, scratch1.stride, p.rw, p.rh));
#undef RED4
        cur = scratch1; cw = p.rw; ch_ = p.rh;
    }
    const char *tb = reinterpret_cast<const char *>(d_tables);
    auto tptr = [&](size_t off) { return reinterpret_cast<const int32_t *>(tb + off); };
    const int32_t *d_bh = tptr(p.off_bh), *d_kh = tptr(p.off_kh), *d_bv = tptr(p.off_bv), *d_kv = tptr(p.off_kv);
    const int32_t *d_khT = tptr(p.off_khT);
    // (the fused kernel addresses a page with 32-bit lane offsets)
    if (p.mm_ok && p.fuse_rv && (size_t)ch_ * cur.pitch < ((size_t)1 << 32) && (size_t)p.oh * dst.pitch < ((size_t)1 << 32)) {
        const ThumbPlan::Mm &H = p.mmh, &Vt = p.mmv;
        const int pw = H.panel_w8;
        int pws = pw + 16;
        if (((pws >> 4) & 1) == 0) pws += 16;
        const int rv = p.fuse_rv;
        const int rmwords = cdiv(ch_, 32);
        const size_t lds = (size_t)2 * 16 * pws + (size_t)128 * 272 + (size_t)rmwords * 4;
        const int gx = cdiv(H.ntiles, 8), nblk = cdiv(Vt.ntiles, rv);
        // blocks per workgroup: as long a run of lines as still leaves some twenty rounds of workgroups (768 run at a time)
        static const int nb_env = getenv("MRCHIP_FUSE_NB") ? atoi(getenv("MRCHIP_FUSE_NB")) : 0;
        int nb = 1;
        while (nb < nblk && (size_t)gx * cdiv(nblk, nb * 2) * npages >= 16384) nb *= 2;
        if (nb_env > 0) nb = nb_env;
        const int gy = cdiv(nblk, nb);
        const dim3 grid(round_up(gx * gy * npages, 8));
        const v4i *bth = reinterpret_cast<const v4i *>(tb + p.off_mm[0][2]), *btv = reinterpret_cast<const v4i *>(tb + p.off_mm[1][2]);
#define MMF_LAUNCH(KH, KV, RVV)                                                                                              \
    LAUNCH(ctx, s, "thumb_resize", red ? 0.0 : alg,                                                                          \
           hipLaunchKernelGGL((resize_mm_fused_kernel<KH, KV, RVV>), grid, dim3(512), lds, s, cur.p, cur.pitch, cur.stride,  \
                              ch_, cw * c, dst.p, dst.pitch, dst.stride, H.nout, H.ntiles, tptr(p.off_mm[0][0]),             \
                              tptr(p.off_mm[0][1]), bth, Vt.nout, Vt.ntiles, tptr(p.off_mm[1][0]), tptr(p.off_mmend[1]),     \
                              tptr(p.off_mm[1][1]), btv, gx, gy, npages, pw, pws, nb, ta ? ta->alt.p : nullptr,              \
                              ta ? ta->alt.pitch : 0, ta ? ta->alt.stride : (size_t)0, ta ? ta->rowmap : nullptr, rmwords))
#define MMF_RV(KH, KV) do { if (rv == 4) MMF_LAUNCH(KH, KV, 4); else if (rv == 2) MMF_LAUNCH(KH, KV, 2); else MMF_LAUNCH(KH, KV, 1); } while (0)
        if (H.KB == 1 && Vt.KB == 1) MMF_RV(1, 1);
        else if (H.KB == 1) MMF_RV(1, 2);
        else if (Vt.KB == 1) MMF_RV(2, 1);
        else MMF_RV(2, 2);
#undef MMF_RV
#undef MMF_LAUNCH
        return 0;
    }
    if (p.mm_ok) {
        // horizontal: lines = image rows -> scratch2 transposed [output byte][row]; vertical: lines = those -> dst
        // (quads of 64 lines per wave: chosen per pass below)
        const ThumbPlan::Mm *mm[2] = {&p.mmh, &p.mmv};
        for (int pass = 0; pass < 2; pass++) {
            const ThumbPlan::Mm &M = *mm[pass];
            const Plane in = pass == 0 ? cur : scratch2, out = pass == 0 ? scratch2 : dst;
            const int nlines = pass == 0 ? ch_ : p.mmh.nout;
            // lines per wave: the horizontal pass (image rows are its lines) likes longer runs than the vertical one
            const int qpw = npages >= 16 ? (pass == 0 ? 16 : 8) : 2;
            constexpr int wpb = 16;
            constexpr int no_panel = 0;
            // 8 tiles per workgroup where their panel fits the 512 loader lanes (16 lines x 32 chunks of 16 bytes): two
            // independent workgroups per CU instead of one of 16 waves (the kernel needs ~82 registers: 4-5 waves per
            // SIMD either way) overlap their barrier / load / MFMA phases -- 128 pages 1.64 + 0.58 -> 1.41 + 0.55 ms
            const bool panel8 = !no_panel && M.kalign == 16 && M.panel_w8 <= 512;
            const bool panel = panel8 || (!no_panel && M.kalign == 16 && M.panel_w <= 1024);
            const int nw = panel8 ? 8 : PNW;
            const int wg_tiles = panel ? nw : wpb;
            const int gx = cdiv(M.ntiles, wg_tiles), gy = cdiv(cdiv(nlines, 64), qpw);
            const dim3 grid(round_up(gx * gy * npages, 8));
            const double a = (pass == 0 && !red) ? alg : 0.0;
            const char *nm = pass == 0 ? "thumb_resize_h" : "thumb_resize_v";
            const v4i *bt = reinterpret_cast<const v4i *>(tb + p.off_mm[pass][2]);
            if (panel) {
                const int pw = panel8 ? M.panel_w8 : M.panel_w;
                int pws = pw + 16;
                if (((pws >> 4) & 1) == 0) pws += 16;          // odd number of 16-byte units per row: conflict-free operand reads
                const size_t lds = (size_t)2 * 16 * pws + (size_t)nw * 16 * 80;
#define MMP_LAUNCH(KBB, NWW)                                                                                              \
    LAUNCH(ctx, s, nm, a,                                                                                                 \
           hipLaunchKernelGGL((resize_mm_panel_kernel<KBB, NWW>), grid, dim3(64 * NWW), lds, s, in.p, in.pitch, in.stride, nlines, \
                              out.p, out.pitch, out.stride, M.nout, M.ntiles, tptr(p.off_mm[pass][0]),                     \
                              tptr(p.off_mm[pass][1]), bt, qpw, pass == 0 ? 1 : 0, gx, gy, npages, pw, pws,            \
                              pass == 0 ? cw * c : ch_))
                if (panel8) { if (M.KB == 1) MMP_LAUNCH(1, 8); else MMP_LAUNCH(2, 8); }
                else { if (M.KB == 1) MMP_LAUNCH(1, 16); else MMP_LAUNCH(2, 16); }
#undef MMP_LAUNCH
                continue;
            }
#define MM_LAUNCH(KBB, WW)                                                                                             \
    LAUNCH(ctx, s, nm, a,                                                                                              \
           hipLaunchKernelGGL((resize_mm_kernel<KBB, WW>), grid, dim3(64 * WW), 0, s, in.p, in.pitch, in.stride, nlines, \
                              out.p, out.pitch, out.stride, M.nout, M.ntiles, tptr(p.off_mm[pass][0]),                  \
                              tptr(p.off_mm[pass][1]), bt, qpw, pass == 0 ? 1 : 0, gx, gy, npages))
            if (M.KB == 1) MM_LAUNCH(1, 16); else MM_LAUNCH(2, 16);
#undef MM_LAUNCH
        }
        return 0;
    }
    if (p.need_h) {
        Plane o = p.need_v ? scratch2 : dst;
        const dim3 grid(cdiv(p.ow, 256), ch_, npages);
        const double a = red ? 0.0 : alg;
        if (!p.khT_.empty()) {
            const dim3 gridr(cdiv(p.ow, 256), cdiv(ch_, RH), npages);
            // widest byte span a 256-column workgroup needs (host tables): LDS staging if it fits a row buffer
            int span = 0;
            for (int x0 = 0; x0 < p.ow; x0 += 256) {
                const int xl = std::min(x0 + 255, p.ow - 1);
                span = std::max(span, (p.bh_[2 * xl] + THUMB_MAXK - p.bh_[2 * x0]) * c);
            }
            const bool use_lds = span + 16 <= RH_LROW;
#define RSZ_H(CC, MK)                                                                                          \
    do {                                                                                                        \
        if (use_lds)                                                                                            \
            LAUNCH(ctx, s, "thumb_resize_h", a,                                                                  \
                   hipLaunchKernelGGL((resize_h_lds_kernel<CC, MK>), gridr, dim3(256), 0, s, cur.p, cur.pitch,   \
                                      cur.stride, ch_, o.p, o.pitch, o.stride, p.ow, d_bh, d_khT));             \
        else                                                                                                    \
            LAUNCH(ctx, s, "thumb_resize_h", a,                                                                  \
                   hipLaunchKernelGGL((resize_h_kernel<CC, MK>), gridr, dim3(256), 0, s, cur.p, cur.pitch,       \
                                      cur.stride, ch_, o.p, o.pitch, o.stride, p.ow, d_bh, d_khT));             \
    } while (0)
            if (c == 3) {
                if (p.ksh <= 9) RSZ_H(3, 9); else if (p.ksh <= 13) RSZ_H(3, 13); else if (p.ksh <= 16) RSZ_H(3, 16); else RSZ_H(3, THUMB_MAXK);
            } else {
                if (p.ksh <= 9) RSZ_H(1, 9); else if (p.ksh <= 13) RSZ_H(1, 13); else if (p.ksh <= 16) RSZ_H(1, 16); else RSZ_H(1, THUMB_MAXK);
            }
#undef RSZ_H
        } else {
            if (c == 3)
                LAUNCH(ctx, s, "thumb_resize_h", a,
                       hipLaunchKernelGGL(resize_h_generic_kernel<3>, grid, dim3(256), 0, s, cur.p, cur.pitch, cur.stride, ch_,
                                          o.p, o.pitch, o.stride, p.ow, d_bh, d_kh, p.ksh));
            else
                LAUNCH(ctx, s, "thumb_resize_h", a,
                       hipLaunchKernelGGL(resize_h_generic_kernel<1>, grid, dim3(256), 0, s, cur.p, cur.pitch, cur.stride, ch_,
                                          o.p, o.pitch, o.stride, p.ow, d_bh, d_kh, p.ksh));
        }
        cur = o; cw = p.ow;
    }
    if (p.need_v) {
        LAUNCH(ctx, s, "thumb_resize_v", (red || p.need_h) ? 0.0 : alg,
               hipLaunchKernelGGL(resize_v_kernel, dim3(cdiv(cdiv(cw * c, 4), 256), p.oh, npages), dim3(256), 0, s, cur.p, cur.pitch,
                                  cur.stride, cw * c, dst.p, dst.pitch, dst.stride, p.oh, d_bv, d_kv, p.ksv));
    } else if (cur.p != dst.p) {
        for (int i = 0; i < npages; i++)
            HIP_TRY(hipMemcpy2DAsync(dst.page(i), dst.pitch, cur.page(i), cur.pitch, (size_t)cw * c, ch_,
                                     hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

}  // namespace mrchip
