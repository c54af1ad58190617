// Gaussian pre-blur (reference: mrc.py:309-311 -> scipy.ndimage.gaussian_filter on
// float32(gray), then .astype(uint8) at mrc.py:325).
//
// scipy's correlate1d with a symmetric kernel, per output sample
//     acc = x[c]*w[r];  for j = -r..-1: acc += (x[c+j] + x[c-j]) * w[r+j]
// in float64, 'reflect' borders (d c b a | a b c d | d c b a), axis 0 first, the
// intermediate rounded to float32, then axis 1; the result is truncated to uint8.
// The weight table is host data (scipy builds it with numpy) and travels as a
// kernel argument (scalar loads).  No FMA contraction anywhere.
//
// v1: two streaming passes (u8 -> f32 tmp -> u8).  Algorithmic bytes 2*w*h.
#include "mrchip_internal.h"

namespace mrchip {


__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i >= 0 && i < n) return i;
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// the same for -n <= i < 2n (one reflection at most): what the fused kernel's tiles can ask for (n >= 2 * GF_RMAX,
// offsets within GF_RMAX of the image) -- two compares and selects instead of the integer modulo, which the compiler
// hoists in front of the interior / border branch and every lane then pays (it was a seventh of the kernel's VALU work)
__device__ __forceinline__ int reflect_once(int i, int n) { return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i); }

// vertical pass: one lane per 4 adjacent pixels (aligned dword loads, float4 store)
__global__ __launch_bounds__(256) void gauss_v_kernel(const uint8_t *src, int spitch, size_t sstride, float *tmp,
                                                      int tpitch, size_t tstride, int w, int h, const GaussW *Gs) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w) return;
    src += (size_t)blockIdx.z * sstride;
    tmp += (size_t)blockIdx.z * tstride;
    const GaussW &G = Gs[blockIdx.z];
    const int r = G.radius;
    auto ld4 = [&](int yy) { return *reinterpret_cast<const unsigned *>(src + (size_t)yy * spitch + x); };
    const unsigned c = ld4(y);
    const double wc = G.w[r];
    double a0 = __dmul_rn((double)(c & 0xffu), wc), a1 = __dmul_rn((double)((c >> 8) & 0xffu), wc);
    double a2 = __dmul_rn((double)((c >> 16) & 0xffu), wc), a3 = __dmul_rn((double)(c >> 24), wc);
    for (int j = -r; j < 0; j++) {
        const unsigned p = ld4(reflect_idx(y + j, h)), q = ld4(reflect_idx(y - j, h));
        const double wj = G.w[r + j];
        a0 = __dadd_rn(a0, __dmul_rn(__dadd_rn((double)(p & 0xffu), (double)(q & 0xffu)), wj));
        a1 = __dadd_rn(a1, __dmul_rn(__dadd_rn((double)((p >> 8) & 0xffu), (double)((q >> 8) & 0xffu)), wj));
        a2 = __dadd_rn(a2, __dmul_rn(__dadd_rn((double)((p >> 16) & 0xffu), (double)((q >> 16) & 0xffu)), wj));
        a3 = __dadd_rn(a3, __dmul_rn(__dadd_rn((double)(p >> 24), (double)(q >> 24)), wj));
    }
    // columns >= w of the last dword are written too: they live in the scratch row's padding
    *reinterpret_cast<float4 *>(tmp + (size_t)y * tpitch + x) = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
}

// horizontal pass: one lane per output (consecutive lanes read consecutive floats)
__global__ __launch_bounds__(256) void gauss_h_kernel(const float *tmp, int tpitch, size_t tstride, uint8_t *dst,
                                                      int dpitch, size_t dstride, int w, int h, const GaussW *Gs) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    tmp += (size_t)blockIdx.z * tstride;
    dst += (size_t)blockIdx.z * dstride;
    const GaussW &G = Gs[blockIdx.z];
    const int r = G.radius;
    const float *row = tmp + (size_t)y * tpitch;
    double acc = __dmul_rn((double)row[x], G.w[r]);
    if (x - r >= 0 && x + r < w) {
        for (int j = -r; j < 0; j++)
            acc = __dadd_rn(acc, __dmul_rn(__dadd_rn((double)row[x + j], (double)row[x - j]), G.w[r + j]));
    } else {
        for (int j = -r; j < 0; j++)
            acc = __dadd_rn(acc, __dmul_rn(__dadd_rn((double)row[reflect_idx(x + j, w)], (double)row[reflect_idx(x - j, w)]),
                                           G.w[r + j]));
    }
    dst[(size_t)y * dpitch + x] = (uint8_t)(float)acc;      // float32 result, astype(uint8) truncation
}

// Fused separable blur for the radii the pipeline produces (sigma_est*0.1 -> radius <= 8): one
// workgroup blurs a 256 x 32 tile.  Vertical pass straight from global memory into a float32 LDS
// tile that includes the 2R halo columns: lane = (4 adjacent columns, 8 consecutive rows), so the
// 8+2R aligned dword loads a lane needs are all issued up front (one exposed memory latency per
// lane, not per output).  Horizontal pass out of LDS (consecutive lanes -> consecutive floats,
// conflict-free); result bytes staged in LDS and written as whole dwords.  Same arithmetic and
// rounding points as the two-pass kernels (float32 intermediate between the passes), but the f32
// scratch image never goes to HBM: 2 B/px of traffic instead of 10.
// R is the largest radius in the batch; a page with a smaller radius has its table zero-padded to
// R by the host: the extra outer taps add +0.0 first, which leaves every partial sum unchanged.
// 240 columns: the tile plus its halo is at most 64 dword groups (4 * 64 >= 240 + 2 * 8), so the vertical pass fills
// exactly the four waves the horizontal pass uses (256 + halo needed a fifth wave with 20 live lanes)
constexpr int GF_TW = 240, GF_TH = 16, GF_RMAX = 8, GF_LW = GF_TW + 2 * GF_RMAX + 4, GF_SEG = 4, GF_THREADS = 256, GF_GROUPS = 64;

template <int R>
__global__ __launch_bounds__(GF_THREADS) void gauss_fused_kernel(const uint8_t *src, int spitch, size_t sstride,
                                                                 uint8_t *dst, int dpitch, size_t dstride, int w, int h,
                                                                 const GaussW *Gs) {
    __shared__ __attribute__((aligned(16))) float tmpT[GF_TH][GF_LW];   // GF_LW % 4 == 0: rows stay 16-B aligned
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const GaussW &G = Gs[blockIdx.z];              // padded to radius R by the host
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * GF_TW, Y0 = blockIdx.y * GF_TH;
    const int Xa = max(0, X0 - R) & ~3;                       // first tile column, dword aligned
    const int Xe = min(w, X0 + GF_TW + R);                    // one past the last needed column
    const int ngroups = (Xe - Xa + 3) >> 2;                   // <= 64
    const int nrows = min(GF_TH, h - Y0);
    double wt[R + 1];                                          // wt[k] = weight of offset -(R-k) (and +(R-k))
#pragma unroll
    for (int k = 0; k <= R; k++) wt[k] = G.w[k];
    // ---- vertical pass: global -> float32 LDS tile ----
    {
        const int g = tid % GF_GROUPS, sgm = tid / GF_GROUPS;  // 64 column groups x 4 row segments
        if (g < ngroups && sgm < GF_TH / GF_SEG) {
            const int x = Xa + 4 * g, y0 = Y0 + sgm * GF_SEG;
            unsigned in[GF_SEG + 2 * R];
            if (Y0 - R >= 0 && Y0 + GF_TH + R <= h) {
                // tile away from the top / bottom border (workgroup-uniform): consecutive rows, no reflection
                const uint8_t *p = src + (size_t)(y0 - R) * spitch + x;
#pragma unroll
                for (int i = 0; i < GF_SEG + 2 * R; i++) in[i] = *reinterpret_cast<const unsigned *>(p + (size_t)i * spitch);
            } else {
#pragma unroll
                for (int i = 0; i < GF_SEG + 2 * R; i++) {
                    const int yy = reflect_once(min(y0 - R + i, h - 1 + R), h);   // rows past the tile's last row are unused
                    in[i] = *reinterpret_cast<const unsigned *>(src + (size_t)yy * spitch + x);
                }
            }
#pragma unroll
            for (int k = 0; k < GF_SEG; k++) {
                if (y0 + k < h) {
                    const unsigned c = in[k + R];
                    double a0 = __dmul_rn((double)(c & 0xffu), wt[R]), a1 = __dmul_rn((double)((c >> 8) & 0xffu), wt[R]);
                    double a2 = __dmul_rn((double)((c >> 16) & 0xffu), wt[R]), a3 = __dmul_rn((double)(c >> 24), wt[R]);
#pragma unroll
                    for (int j = R; j >= 1; j--) {            // offsets -j and +j, outermost first (scipy order)
                        const unsigned p = in[k + R - j], q = in[k + R + j];
                        const double wj = wt[R - j];
                        // (double)a + (double)b of two bytes == (double)(a + b): one conversion
                        a0 = __dadd_rn(a0, __dmul_rn((double)((p & 0xffu) + (q & 0xffu)), wj));
                        a1 = __dadd_rn(a1, __dmul_rn((double)(((p >> 8) & 0xffu) + ((q >> 8) & 0xffu)), wj));
                        a2 = __dadd_rn(a2, __dmul_rn((double)(((p >> 16) & 0xffu) + ((q >> 16) & 0xffu)), wj));
                        a3 = __dadd_rn(a3, __dmul_rn((double)((p >> 24) + (q >> 24)), wj));
                    }
                    // one 16-byte LDS store (four dword stores 16 B apart per lane would be a 4-way bank conflict)
                    *reinterpret_cast<float4 *>(&tmpT[sgm * GF_SEG + k][4 * g]) = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
                }
            }
        }
    }
    __syncthreads();
    // ---- horizontal pass: LDS -> global.  lane = (4 adjacent columns, 8 consecutive rows): the 4 + 2R floats a lane
    // needs come in as 16-byte LDS reads and are widened to double once each; the four result bytes leave as one dword
    {
        constexpr int RP = (R <= 4) ? 4 : 8;                      // halo rounded up to whole float4s
        const int q = tid & 63, sgm = tid >> 6;
        const int x0 = X0 + 4 * q;
        if (q < GF_TW / 4 && x0 < w) {
            const bool interior = (x0 - R >= 0) && (x0 + 3 + R < w);
            // the taps of one output row of this lane: v[i] = intermediate of column x0 - RP + i
            auto finish_row = [&](int ty, const double (&v)[4 + 2 * RP]) {
                unsigned packed = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    double acc = __dmul_rn(v[RP + e], wt[R]);
#pragma unroll
                    for (int j = R; j >= 1; j--)
                        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(v[RP + e - j], v[RP + e + j]), wt[R - j]));
                    packed |= (unsigned)(uint8_t)(float)acc << (8 * e);   // float32 result, astype(uint8) truncation
                }
                uint8_t *o = dst + (size_t)(Y0 + ty) * dpitch + x0;
                if (x0 + 4 <= w) *reinterpret_cast<unsigned *>(o) = packed;
                else for (int i = 0; x0 + i < w; i++) o[i] = (uint8_t)(packed >> (8 * i));
            };
            // Waves whose lanes all sit inside the image (every wave of a tile away from the left / right border) take
            // 16-byte LDS reads; decided per WAVE: with a per-lane branch the compiler merges the two paths into
            // per-element reads with selected indices.
            if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
                // x0 - RP - Xa is a multiple of 4 (X0 - Xa is 0 or RP, x0 - X0 = 4q)
                const int off = (x0 - RP - Xa) & ~3;
#pragma unroll 2
                for (int k = 0; k < GF_SEG; k++) {
                    const int ty = sgm * GF_SEG + k;
                    if (ty >= nrows) break;
                    const float *seg = static_cast<const float *>(__builtin_assume_aligned(&tmpT[ty][off], 16));
                    double v[4 + 2 * RP];
#pragma unroll
                    for (int i = 0; i < (4 + 2 * RP) / 4; i++) {
                        const float4 f = *reinterpret_cast<const float4 *>(seg + 4 * i);
                        v[4 * i] = (double)f.x; v[4 * i + 1] = (double)f.y; v[4 * i + 2] = (double)f.z; v[4 * i + 3] = (double)f.w;
                    }
                    finish_row(ty, v);
                }
            } else {
                for (int k = 0; k < GF_SEG; k++) {
                    const int ty = sgm * GF_SEG + k;
                    if (ty >= nrows) break;
                    const float *row = &tmpT[ty][0] - Xa;             // row[c] = intermediate of image column c
                    double v[4 + 2 * RP];
#pragma unroll
                    for (int i = 0; i < 4 + 2 * RP; i++) v[i] = 0.0;
#pragma unroll
                    for (int i = RP - R; i < 4 + RP + R; i++) v[i] = (double)row[reflect_once(x0 - RP + i, w)];
                    finish_row(ty, v);
                }
            }
        }
    }
}

// true when launch_gaussian_batch will take the fused kernel, i.e. the tables must be padded
bool gauss_uses_fused(int w, int h, int max_radius) {
    return max_radius >= 1 && max_radius <= GF_RMAX && w >= 2 * GF_RMAX && h >= 2 * GF_RMAX;
}

// re-centre a radius-r table inside a radius-R one (zeros outside): taps added as +0.0
void gauss_pad_weights(GaussW &g, int R) {
    const int r = g.radius;
    if (r >= R) return;
    double t[2 * GMAXR + 1];
    for (int i = 0; i < 2 * r + 1; i++) t[i] = g.w[i];
    for (int i = 0; i < 2 * R + 1; i++) g.w[i] = 0.0;
    for (int i = 0; i < 2 * r + 1; i++) g.w[i + (R - r)] = t[i];
    g.radius = R;
}

int gaussian_weights_libm(double sigma, std::vector<double> &wts) {
    // scipy _gaussian_kernel1d with libm's exp (numpy's exp may differ in the last bit)
    int radius = (int)(4.0 * sigma + 0.5);
    if (radius > GMAXR) { set_error("gaussian: radius %d > %d", radius, GMAXR); return MRCHIP_E_UNSUPPORTED; }
    wts.assign(2 * radius + 1, 0.0);
    double c = -0.5 / (sigma * sigma);
    for (int i = -radius; i <= radius; i++) wts[i + radius] = exp(c * (double)(i * i));
    // numpy pairwise summation order
    const int n = 2 * radius + 1;
    double s;
    if (n < 8) { s = 0; for (int i = 0; i < n; i++) s += wts[i]; }
    else {
        double r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = wts[i];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += wts[i + j];
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) s += wts[i];
    }
    for (int i = 0; i < n; i++) wts[i] = wts[i] / s;
    return 0;
}

// radius 0 (weight 1.0) is the identity: float32(u8) -> u8, so pages without blur (sigma_est <= 1,
// mrc.py:309) ride along in the same launch.
int launch_gaussian_batch(mrchip_ctx *ctx, hipStream_t s, Plane src, Plane dst, int w, int h, const GaussW *d_weights,
                          float *tmp, int tpitch, size_t tstride, int npages, int max_radius) {
    if (max_radius >= 1 && max_radius <= GF_RMAX && w >= 2 * GF_RMAX && h >= 2 * GF_RMAX) {
        // d_weights must be padded to max_radius (gauss_pad_weights)
        dim3 gridf(cdiv(w, GF_TW), cdiv(h, GF_TH), npages);
#define GF_CASE(RR)                                                                                              \
    case RR:                                                                                                     \
        LAUNCH(ctx, s, "gauss_fused", 2.0 * w * h * npages,                                                      \
               hipLaunchKernelGGL(gauss_fused_kernel<RR>, gridf, dim3(GF_THREADS), 0, s, src.p, src.pitch, src.stride, \
                                  dst.p, dst.pitch, dst.stride, w, h, d_weights));                                \
        break;
        switch (max_radius) { GF_CASE(1) GF_CASE(2) GF_CASE(3) GF_CASE(4) GF_CASE(5) GF_CASE(6) GF_CASE(7) GF_CASE(8) }
#undef GF_CASE
        return 0;
    }
    dim3 grid_v(cdiv(cdiv(w, 4), 256), h, npages);
    dim3 grid(cdiv(w, 256), h, npages);
    LAUNCH(ctx, s, "gauss_v", 5.0 * w * h * npages,
           hipLaunchKernelGGL(gauss_v_kernel, grid_v, dim3(256), 0, s, src.p, src.pitch, src.stride, tmp, tpitch, tstride, w, h,
                              d_weights));
    LAUNCH(ctx, s, "gauss_h", 5.0 * w * h * npages,
           hipLaunchKernelGGL(gauss_h_kernel, grid, dim3(256), 0, s, tmp, tpitch, tstride, dst.p, dst.pitch, dst.stride, w, h,
                              d_weights));
    return 0;
}

}  // namespace mrchip
