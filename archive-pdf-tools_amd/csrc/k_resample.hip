// PIL Image.thumbnail((int(w/f), int(h/f))) (reference: mrc.py:422-428, 456-462; Pillow
// Image.thumbnail -> Image.reduce (Reduce.c box mean) -> Image.resize BICUBIC (Resample.c,
// 8 bpc fixed point, PRECISION_BITS 22, horizontal pass then vertical pass); SURVEY.md 8a
// row a11).  The size rule and the coefficient tables are host logic (double precision,
// exactly as Pillow computes them); the pixel passes are integer HIP kernels.
// Algorithmic bytes: C*P*(1 + 1/f^2) per layer.
#include <cmath>

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

// Resample.c precompute_coeffs + normalize_coeffs_8bpc; box edges are float32 like Pillow's
int bicubic_coeffs(int in_size, float in0, float in1, int out_size, std::vector<int32_t> &bounds,
                   std::vector<int32_t> &kk) {
    double scale = (double)(in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = 2.0 * filterscale;
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
            double wv = bicubic_filter((x + xmin - center + 0.5) * ss);
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

// Plan of one thumbnail: host tables + scratch sizes.  Built once per (shape, request).
int ThumbPlan_build(ThumbPlan &p, int w, int h, int c, int req_w, int req_h) {
    p = ThumbPlan();
    p.w = w; p.h = h; p.c = c;
    p.changed = thumbnail_size(w, h, req_w, req_h, &p.ow, &p.oh);
    if (!p.changed) return 0;
    // Image.resize: factor = int(extent / size / reducing_gap) or 1   (reducing_gap = 2.0)
    p.fx = (int)((double)w / p.ow / 2.0); if (p.fx < 1) p.fx = 1;
    p.fy = (int)((double)h / p.oh / 2.0); if (p.fy < 1) p.fy = 1;
    float bw = (float)w, bh = (float)h;
    p.rw = w; p.rh = h;
    if (p.fx > 1 || p.fy > 1) {
        p.rw = (w + p.fx - 1) / p.fx; p.rh = (h + p.fy - 1) / p.fy;
        bw = (float)((double)w / p.fx); bh = (float)((double)h / p.fy);
    }
    p.need_h = (p.ow != p.rw) || (bw != (float)p.ow);
    p.need_v = (p.oh != p.rh) || (bh != (float)p.oh);
    if (p.need_h) {
        p.ksh = bicubic_coeffs(p.rw, 0.f, bw, p.ow, p.bh_, p.kh_);
        if (p.ksh <= THUMB_MAXK) {                 // transposed, zero-padded to THUMB_MAXK taps
            p.khT_.assign((size_t)THUMB_MAXK * p.ow, 0);
            for (int xx = 0; xx < p.ow; xx++)
                for (int x = 0; x < p.ksh; x++) p.khT_[(size_t)x * p.ow + xx] = p.kh_[(size_t)xx * p.ksh + x];
        }
    }
    if (p.need_v) p.ksv = bicubic_coeffs(p.rh, 0.f, bh, p.oh, p.bv_, p.kv_);
    return 0;
}

size_t ThumbPlan_table_bytes(const ThumbPlan &p) {
    return (p.bh_.size() + p.kh_.size() + p.bv_.size() + p.kv_.size() + p.khT_.size()) * sizeof(int32_t) + 64;
}

// d_tables: device copy of [bh_, kh_, bv_, kv_] in that order (int32), made by the caller.
// scratch1: rw*rh*c per page (reduce output, if any); scratch2: ow*rh*c per page (horizontal pass output)
int launch_thumbnail_plan(mrchip_ctx *ctx, hipStream_t s, const ThumbPlan &p, Plane src, Plane dst,
                          const int32_t *d_tables, Plane scratch1, Plane scratch2, int npages) {
    const int c = p.c;
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
        LAUNCH(ctx, s, "thumb_reduce", alg,
               hipLaunchKernelGGL(reduce_kernel, dim3(cdiv(p.rw, 256), p.rh, npages), dim3(256), 0, s, cur.p, cur.pitch,
                                  cur.stride, cw, ch_, c, p.fx, p.fy, scratch1.p, scratch1.pitch, scratch1.stride, p.rw, p.rh));
        cur = scratch1; cw = p.rw; ch_ = p.rh;
    }
    const int32_t *d_bh = d_tables, *d_kh = d_bh + p.bh_.size(), *d_bv = d_kh + p.kh_.size(), *d_kv = d_bv + p.bv_.size();
    const int32_t *d_khT = d_kv + p.kv_.size();
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
