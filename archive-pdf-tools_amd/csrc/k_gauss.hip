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

// vertical pass over a float32 source (round 6: mrc.create_threshold_mask on a float32 image that does not hold whole
// numbers 0..255): scipy's line buffer is double, so the pair sum is a rounded float64 addition like every other step
__global__ __launch_bounds__(256) void gauss_v_f32_kernel(const float *src, int spitch, float *tmp, int tpitch, int w, int h,
                                                          const GaussW *Gs) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const GaussW &G = Gs[0];
    const int r = G.radius;
    double acc = __dmul_rn((double)src[(size_t)y * spitch + x], G.w[r]);
    for (int j = -r; j < 0; j++)
        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn((double)src[(size_t)reflect_idx(y + j, h) * spitch + x],
                                                  (double)src[(size_t)reflect_idx(y - j, h) * spitch + x]), G.w[r + j]));
    tmp[(size_t)y * tpitch + x] = (float)acc;
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
// (16 rows: tiles of 32 / 64 rows re-read less of their neighbours' rows -- the counters show 1.87 B read per pixel at 16 --
// but fit fewer workgroups per CU and ran 10 % / 47 % slower)
constexpr int GF_TW = 240, GF_TH = 16, GF_RMAX = 8, GF_LW = GF_TW + 2 * GF_RMAX + 4, GF_SEG = 4, GF_THREADS = 256, GF_GROUPS = 64;

template <int R>
__device__ __forceinline__ void gauss_fused_tile(float (&tmpT)[GF_TH][GF_LW], const uint8_t *src, int spitch, size_t sstride,
                                                 uint8_t *dst, int dpitch, size_t dstride, int w, int h, const GaussW *Gs,
                                                 const int bx, const int by, const int bz) {
    src += (size_t)bz * sstride;
    dst += (size_t)bz * dstride;
    const GaussW &G = Gs[bz];              // padded to radius R by the host
    const int tid = threadIdx.x;
    const int X0 = bx * GF_TW, Y0 = by * GF_TH;
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

template <int R>
__global__ __launch_bounds__(GF_THREADS) void gauss_fused_kernel(const uint8_t *src, int spitch, size_t sstride,
                                                                 uint8_t *dst, int dpitch, size_t dstride, int w, int h,
                                                                 const GaussW *Gs) {
    __shared__ __attribute__((aligned(16))) float tmpT[GF_TH][GF_LW];   // GF_LW % 4 == 0: rows stay 16-B aligned
    gauss_fused_tile<R>(tmpT, src, spitch, sstride, dst, dpitch, dstride, w, h, Gs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- the fast form: both passes in float32, the few pixels it cannot decide in float64 afterwards ------------------
// The reference's result is b = trunc(fl32(A)), A = the float64 horizontal sum over float32 intermediates v = fl32(V),
// V = the float64 vertical sum (scipy: float32 array between the axes, mrc.py:309-311 / 325).  gauss_fast_kernel
// evaluates the same two sums in float32 (round to nearest, one rounding per multiply / add / fma; packed: two pixels per
// instruction).  With u = 2^-17 (half an ulp of a float32 in [128, 256): every partial sum is < 256 because the weights
// are positive and sum to 1) and 255 2^-24 < 2^-16 for the weights' own rounding:
//     vertical:    |v~ - V| <= (R + 1) u + 2^-16   (R + 1 roundings, weights),  |v - V| <= u
//                  =>  |v~ - v| <= e1(R) = (R + 2) u + 2^-16      -- checked EXHAUSTIVELY for R <= 2 over every (centre,
//                                                                     pair sum, pair sum): mrchip_selftest_gauss_fast
//     horizontal:  |a - A| <= (R + 1) u   (roundings of the running sum)  +  u   (pair sums < 512 round by <= 2^-16, times
//                  the one-sided weights' sum <= 1/2)  +  2^-16 (weights)  +  e1 (the inputs' error, passed through by weights
//                  that sum to 1)  =  (2 R + 4) u + 2^-15      -- sampled 2^28 times by the same self-test
//     fl32(A) differs from A by <= u, so a pixel whose a lies further than D(R) = (2 R + 5) u + 2^-15 from every integer
//     has trunc(a) = b.
// The kernel takes GQ_E(R) = (2 R + 12) u  (R = 2: 1.22e-4 against D = 9.9e-5; R = 8: 2.14e-4 against 1.91e-4) and works on
// y = a - 1/2 (the -1/2 rides in the first fma): the byte is v_cvt_pk_u8_f32(y) -- round to nearest even of a - 1/2 = the
// floor away from ties, saturating, dropped into place -- and a pixel is undecided iff |fract(y) - 1/2| < GQ_E.  Those --
// 2 GQ_E = 0.024 % of the pixels of a noisy page at R = 2 -- are listed per tile and recomputed by gauss_fix_kernel with
// the reference's float64 sequence.  A tile with more than GQ_CAP of them (flat areas: a constant window gives an integer)
// is marked and redone whole by the float64 tile code (gauss_exact_tiles_kernel).  Every tile owns GQ_CAP slots and one
// count word (0xffffffff = redo the tile): nothing can overflow, and no launch-wide counter is hammered (a first
// version appended to one global list: 1.5 million same-address atomics per launch cost more than the blur).
constexpr float gq_e(int R) { return (float)(2 * R + 12) / 131072.0f; }
constexpr int GQ_CAP = 32;
constexpr int GQ_INLINE_R = 3;        // radii whose float64 pixel code is small enough to ride inside the fast kernel
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ub(unsigned v, int e) { return (float)((v >> (8 * e)) & 0xffu); }      // v_cvt_f32_ubyteN

// one pixel with the reference's float64 sequence (G padded to radius R).  Away from the image border its (2R+1)^2
// neighbourhood comes in as whole dwords, all issued before the first use.
template <int R>
__device__ __forceinline__ void gauss_fix_pixel(const uint8_t *sp, int spitch, uint8_t *dp, int dpitch, int w, int h, const GaussW &G,
                                                const int x, const int y) {
    double wt[R + 1];
#pragma unroll
    for (int k = 0; k <= R; k++) wt[k] = G.w[k];
    float v[2 * R + 1];
    if (x >= R && x + R < w && y >= R && y + R < h) {
        // rows y-R .. y+R, bytes x-R .. x+R of each: aligned dwords (rows are padded), realigned to the first column
        constexpr int NE = (2 * R + 1 + 3) / 4;         // dwords of 2R+1 bytes
        const int xa = (x - R) & ~3;
        const unsigned sh = (unsigned)((x - R) & 3);
        unsigned d[2 * R + 1][NE + 1];
#pragma unroll
        for (int r = 0; r <= 2 * R; r++) {
            const unsigned *rp = reinterpret_cast<const unsigned *>(sp + (size_t)(y - R + r) * spitch + xa);
#pragma unroll
            for (int k = 0; k <= NE; k++) d[r][k] = rp[k];
        }
        unsigned eb[2 * R + 1][NE];
#pragma unroll
        for (int r = 0; r <= 2 * R; r++)
#pragma unroll
            for (int k = 0; k < NE; k++) eb[r][k] = __builtin_amdgcn_alignbyte(d[r][k + 1], d[r][k], sh);
        auto px = [&](int r, int cidx) { return (eb[r][cidx >> 2] >> (8 * (cidx & 3))) & 0xffu; };
#pragma unroll
        for (int cidx = 0; cidx <= 2 * R; cidx++) {
            double a = __dmul_rn((double)px(R, cidx), wt[R]);
#pragma unroll
            for (int j = R; j >= 1; j--) a = __dadd_rn(a, __dmul_rn((double)(px(R - j, cidx) + px(R + j, cidx)), wt[R - j]));
            v[cidx] = (float)a;
        }
    } else {
#pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            const uint8_t *col = sp + reflect_once(x + dx, w);          // (w, h >= 2 * GF_RMAX: one reflection at most)
            double a = __dmul_rn((double)col[(size_t)y * spitch], wt[R]);
#pragma unroll
            for (int j = R; j >= 1; j--) {
                const unsigned p = col[(size_t)reflect_once(y - j, h) * spitch], q = col[(size_t)reflect_once(y + j, h) * spitch];
                a = __dadd_rn(a, __dmul_rn((double)(p + q), wt[R - j]));
            }
            v[dx + R] = (float)a;
        }
    }
    double acc = __dmul_rn((double)v[R], wt[R]);
#pragma unroll
    for (int j = R; j >= 1; j--) acc = __dadd_rn(acc, __dmul_rn(__dadd_rn((double)v[R - j], (double)v[R + j]), wt[R - j]));
    dp[(size_t)y * dpitch + x] = (uint8_t)(float)acc;
}

template <int R>
__global__ __launch_bounds__(GF_THREADS) void gauss_fast_kernel(const uint8_t *src, int spitch, size_t sstride,
                                                                uint8_t *dst, int dpitch, size_t dstride, int w, int h,
                                                                const GaussW *Gs, unsigned *tcnt, unsigned *tslots) {
    __shared__ __attribute__((aligned(16))) float tmpT[GF_TH][GF_LW];
    __shared__ unsigned lst[GQ_CAP];
    __shared__ unsigned lcnt;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    const GaussW &G = Gs[blockIdx.z];              // padded to radius R by the host
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * GF_TW, Y0 = blockIdx.y * GF_TH;
    const int Xa = max(0, X0 - R) & ~3;
    const int Xe = min(w, X0 + GF_TW + R);
    const int ngroups = (Xe - Xa + 3) >> 2;
    const int nrows = min(GF_TH, h - Y0);
    const unsigned tile_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid == 0) lcnt = 0;
    if (G.w[R] == 1.0) {
        if (tid == 0) tcnt[tile_id] = 0;
        // a page that is not blurred (radius 0, padded): float32(u8) -> u8 is the identity.  (In float32 every pixel
        // would come out an exact integer, i.e. undecided.)
        for (int i = tid; i < nrows * (GF_TW / 4); i += GF_THREADS) {
            const int ty = i / (GF_TW / 4), x0 = X0 + 4 * (i % (GF_TW / 4));
            if (x0 >= w) continue;
            const unsigned v = *reinterpret_cast<const unsigned *>(src + (size_t)(Y0 + ty) * spitch + x0);
            uint8_t *o = dst + (size_t)(Y0 + ty) * dpitch + x0;
            if (x0 + 4 <= w) *reinterpret_cast<unsigned *>(o) = v;
            else for (int k = 0; x0 + k < w; k++) o[k] = (uint8_t)(v >> (8 * k));
        }
        return;
    }
    float wt[R + 1];
#pragma unroll
    for (int k = 0; k <= R; k++) wt[k] = (float)G.w[k];
    // ---- vertical pass in float32: global -> LDS tile (same lane mapping as the float64 tile code) ----
    {
        const int g = tid % GF_GROUPS, sgm = tid / GF_GROUPS;
        if (g < ngroups && sgm < GF_TH / GF_SEG) {
            const int x = Xa + 4 * g, y0 = Y0 + sgm * GF_SEG;
            unsigned in[GF_SEG + 2 * R];
            if (Y0 - R >= 0 && Y0 + GF_TH + R <= h) {
                const uint8_t *p = src + (size_t)(y0 - R) * spitch + x;
#pragma unroll
                for (int i = 0; i < GF_SEG + 2 * R; i++) in[i] = *reinterpret_cast<const unsigned *>(p + (size_t)i * spitch);
            } else {
#pragma unroll
                for (int i = 0; i < GF_SEG + 2 * R; i++) {
                    const int yy = reflect_once(min(y0 - R + i, h - 1 + R), h);
                    in[i] = *reinterpret_cast<const unsigned *>(src + (size_t)yy * spitch + x);
                }
            }
#pragma unroll
            for (int k = 0; k < GF_SEG; k++) {
                if (y0 + k < h) {
                    const unsigned c = in[k + R];
                    f32x2 a01 = f32x2{ub(c, 0), ub(c, 1)} * wt[R], a23 = f32x2{ub(c, 2), ub(c, 3)} * wt[R];
#pragma unroll
                    for (int j = R; j >= 1; j--) {
                        const unsigned p = in[k + R - j], q = in[k + R + j];
                        const f32x2 s01 = f32x2{ub(p, 0), ub(p, 1)} + f32x2{ub(q, 0), ub(q, 1)};     // exact: integers <= 510
                        const f32x2 s23 = f32x2{ub(p, 2), ub(p, 3)} + f32x2{ub(q, 2), ub(q, 3)};
                        const f32x2 wj = f32x2{wt[R - j], wt[R - j]};
                        a01 = __builtin_elementwise_fma(s01, wj, a01);
                        a23 = __builtin_elementwise_fma(s23, wj, a23);
                    }
                    *reinterpret_cast<float4 *>(&tmpT[sgm * GF_SEG + k][4 * g]) = make_float4(a01.x, a01.y, a23.x, a23.y);
                }
            }
        }
    }
    __syncthreads();
    // ---- horizontal pass in float32: LDS -> bytes; undecided pixels into the workgroup's list ----
    {
        constexpr int RP = (R <= 4) ? 4 : 8;
        const int q = tid & 63, sgm = tid >> 6;
        const int x0 = X0 + 4 * q;
        if (q < GF_TW / 4 && x0 < w) {
            const bool interior = (x0 - R >= 0) && (x0 + 3 + R < w);
            auto finish_row = [&](int ty, const float (&v)[4 + 2 * RP]) {
                const f32x2 mhalf = f32x2{-0.5f, -0.5f}, wc = f32x2{wt[R], wt[R]};
                f32x2 y01 = __builtin_elementwise_fma(f32x2{v[RP], v[RP + 1]}, wc, mhalf);
                f32x2 y23 = __builtin_elementwise_fma(f32x2{v[RP + 2], v[RP + 3]}, wc, mhalf);
#pragma unroll
                for (int j = R; j >= 1; j--) {
                    const f32x2 wj = f32x2{wt[R - j], wt[R - j]};
                    y01 = __builtin_elementwise_fma(f32x2{v[RP - j], v[RP + 1 - j]} + f32x2{v[RP + j], v[RP + 1 + j]}, wj, y01);
                    y23 = __builtin_elementwise_fma(f32x2{v[RP + 2 - j], v[RP + 3 - j]} + f32x2{v[RP + 2 + j], v[RP + 3 + j]}, wj, y23);
                }
                const float y[4] = {y01.x, y01.y, y23.x, y23.y};
                unsigned packed = 0, amb = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    packed = __builtin_amdgcn_cvt_pk_u8_f32(y[e], e, packed);
                    amb |= (__builtin_fabsf(__builtin_amdgcn_fractf(y[e]) - 0.5f) < gq_e(R)) ? (1u << e) : 0u;
                }
                uint8_t *o = dst + (size_t)(Y0 + ty) * dpitch + x0;
                if (x0 + 4 <= w) *reinterpret_cast<unsigned *>(o) = packed;
                else for (int i = 0; x0 + i < w; i++) o[i] = (uint8_t)(packed >> (8 * i));
                if (amb) {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (((amb >> e) & 1u) && x0 + e < w) {
                            const unsigned k = atomicAdd(&lcnt, 1u);
                            if (k < (unsigned)GQ_CAP) lst[k] = ((unsigned)ty << 8) | (unsigned)(4 * q + e);
                        }
                }
            };
            if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
                const int off = (x0 - RP - Xa) & ~3;
#pragma unroll 2
                for (int k = 0; k < GF_SEG; k++) {
                    const int ty = sgm * GF_SEG + k;
                    if (ty >= nrows) break;
                    const float *seg = static_cast<const float *>(__builtin_assume_aligned(&tmpT[ty][off], 16));
                    float v[4 + 2 * RP];
#pragma unroll
                    for (int i = 0; i < (4 + 2 * RP) / 4; i++) {
                        const float4 f = *reinterpret_cast<const float4 *>(seg + 4 * i);
                        v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
                    }
                    finish_row(ty, v);
                }
            } else {
                for (int k = 0; k < GF_SEG; k++) {
                    const int ty = sgm * GF_SEG + k;
                    if (ty >= nrows) break;
                    const float *row = &tmpT[ty][0] - Xa;
                    float v[4 + 2 * RP];
#pragma unroll
                    for (int i = 0; i < 4 + 2 * RP; i++) v[i] = 0.0f;
#pragma unroll
                    for (int i = RP - R; i < 4 + RP + R; i++) v[i] = row[reflect_once(x0 - RP + i, w)];
                    finish_row(ty, v);
                }
            }
        }
    }
    // The inline clean-up below rewrites single bytes of dwords that OTHER waves of this workgroup have just stored.  At
    // workgroup scope the compiler's barrier waits for lgkmcnt only (the listing shows no vmcnt before s_barrier): by the
    // gfx942 / gfx950 memory model the stores of one CU reach memory in issue order, so the patch cannot overtake the
    // dword -- but nothing here has to lean on that: every wave sees its own stores acknowledged before it arrives
    // (+1.8 % on the launch; tests/test_isa_checks.py looks for the wait in the listing).  Tried instead: the undecided
    // groups kept back and stored once by the recomputing lane, +10 %; every wave patching the pixels of its own rows,
    // +10 % (the float64 pixel code then runs in up to four waves per tile instead of one); the wait only in the rare
    // listing branch, +3.8 %.
    if constexpr (R <= GQ_INLINE_R) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned n = lcnt;
    if constexpr (R <= GQ_INLINE_R) {
        // the undecided pixels right here, while the tile's rows are still in the caches (as a pass of its own the scattered
        // 128-byte reads from HBM cost 0.19 ms per 128 pages -- a fifth of the blur): n lanes of the first wave
        if (tid == 0) tcnt[tile_id] = n > (unsigned)GQ_CAP ? 0xffffffffu : 0u;
        if ((unsigned)tid < n && n <= (unsigned)GQ_CAP) {
            const unsigned e = lst[tid];
            // (the byte this lane rewrites was stored by another wave of this workgroup, and acknowledged, before the barrier)
            gauss_fix_pixel<R>(src, spitch, dst, dpitch, w, h, G, X0 + (int)(e & 0xffu), Y0 + (int)(e >> 8));
        }
    } else {
        if (tid == 0) tcnt[tile_id] = n > (unsigned)GQ_CAP ? 0xffffffffu : n;
        if ((unsigned)tid < n && n <= (unsigned)GQ_CAP) tslots[(size_t)tile_id * GQ_CAP + tid] = lst[tid];     // (row << 8) | column inside the tile
    }
}

// The listed pixels with the reference's float64 sequence (the weights table is padded to radius R), one lane each.  A
// tile lists ~4 of its 3840 pixels: a workgroup takes 256 tiles at a time, scans their counts and hands entry i of the
// concatenated lists to lane i mod 256 -- full waves (with a lane per slot 7 of 8 lanes idled).  A pixel away from the
// image border loads its (2R+1)^2 neighbourhood as whole dwords, all issued before the first use (as byte loads in a loop
// over runtime bounds, weights fetched per tap, this clean-up cost 0.7 ms per 128 pages -- half the blur).
template <int R>
__global__ __launch_bounds__(256) void gauss_fix_kernel(const uint8_t *src, int spitch, size_t sstride, uint8_t *dst, int dpitch,
                                                        size_t dstride, int w, int h, const GaussW *Gs, const unsigned *tcnt,
                                                        const unsigned *tslots, unsigned ntiles, int gx, int gy) {
    __shared__ unsigned pre[2][256];
    __shared__ unsigned total;
    const int tid = threadIdx.x;
    for (unsigned chunk = blockIdx.x; chunk * 256u < ntiles; chunk += gridDim.x) {
        const unsigned t0 = chunk * 256u + tid;
        unsigned c = t0 < ntiles ? tcnt[t0] : 0u;
        if (c == 0xffffffffu) c = 0u;                       // a marked tile is redone whole
        // inclusive scan over the 256 counts (Hillis-Steele in LDS, double buffered)
        int cur = 0;
        pre[0][tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const unsigned v = pre[cur][tid] + (tid >= d ? pre[cur][tid - d] : 0u);
            pre[cur ^ 1][tid] = v;
            cur ^= 1;
            __syncthreads();
        }
        if (tid == 255) total = pre[cur][255];
        __syncthreads();
        const unsigned T = total;
        for (unsigned i = tid; i < T; i += 256) {
            // the tile whose entries cover i: the first k with inclusive[k] > i
            int lo = 0, hi = 255;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (pre[cur][mid] > i) hi = mid; else lo = mid + 1;
            }
            const unsigned t = chunk * 256u + (unsigned)lo;
            const unsigned slot = i - (lo ? pre[cur][lo - 1] : 0u);
            const unsigned e = tslots[(size_t)t * GQ_CAP + slot];
            const int bx = (int)(t % (unsigned)gx), by = (int)((t / (unsigned)gx) % (unsigned)gy), page = (int)(t / ((unsigned)gx * (unsigned)gy));
            const int y = by * GF_TH + (int)(e >> 8), x = bx * GF_TW + (int)(e & 0xffu);
            gauss_fix_pixel<R>(src + (size_t)page * sstride, spitch, dst + (size_t)page * dstride, dpitch, w, h, Gs[page], x, y);
        }
        __syncthreads();
    }
}

// the marked tiles with the float64 tile code: every workgroup scans its share of the count words (normally none is marked)
constexpr unsigned GQ_REDO_CHUNK = 32;
template <int R>
__global__ __launch_bounds__(GF_THREADS) void gauss_exact_tiles_kernel(const uint8_t *src, int spitch, size_t sstride,
                                                                       uint8_t *dst, int dpitch, size_t dstride, int w, int h,
                                                                       const GaussW *Gs, const unsigned *tcnt, unsigned ntiles,
                                                                       int gx, int gy) {
    __shared__ __attribute__((aligned(16))) float tmpT[GF_TH][GF_LW];
    // GQ_REDO_CHUNK count words per workgroup and round: marked tiles come in runs (a flat margin is a block of them), and
    // a workgroup redoes the ones it found one after the other -- with 256 words per round and one workgroup per CU
    // (round 5) a page with clipped-white margins left most CUs idle behind a few long queues
    __shared__ unsigned marked[GQ_REDO_CHUNK];
    __shared__ unsigned nmarked;
    for (unsigned base = blockIdx.x * GQ_REDO_CHUNK; base < ntiles; base += gridDim.x * GQ_REDO_CHUNK) {
        if (threadIdx.x == 0) nmarked = 0;
        __syncthreads();
        const unsigned t = base + threadIdx.x;
        if (threadIdx.x < GQ_REDO_CHUNK && t < ntiles && tcnt[t] == 0xffffffffu) marked[atomicAdd(&nmarked, 1u)] = t;
        __syncthreads();
        const unsigned nm = nmarked;
        for (unsigned i = 0; i < nm; i++) {
            const unsigned tt = marked[i];
            gauss_fused_tile<R>(tmpT, src, spitch, sstride, dst, dpitch, dstride, w, h, Gs, (int)(tt % (unsigned)gx),
                                (int)((tt / (unsigned)gx) % (unsigned)gy), (int)(tt / ((unsigned)gx * (unsigned)gy)));
            __syncthreads();
        }
    }
}

// Self-test of the vertical stage's error bound e1 (see above) for a table of radius <= 2: every (centre byte, pair sum,
// pair sum) -- 256 x 511 x 511 -- through the float32 sequence of gauss_fast_kernel and through the float64 one.
__global__ __launch_bounds__(256) void gauss_fast_selftest_kernel(const GaussW *Gs, unsigned long long *bad, unsigned *maxerr_bits) {
    const GaussW &G = Gs[0];
    const int R = G.radius;                      // 1 or 2
    const float w0 = (float)G.w[R], w1 = (float)G.w[R - 1], w2 = R >= 2 ? (float)G.w[R - 2] : 0.0f;
    const float e1 = (float)(R + 2) * (1.0f / 131072.0f) + 255.0f / 16777216.0f;
    const int c = blockIdx.x;                    // centre byte
    unsigned long long nbad = 0;
    float worst = 0.0f;
    for (int s1 = threadIdx.x; s1 <= 510; s1 += 256)
        for (int s2 = 0; s2 <= (R >= 2 ? 510 : 0); s2++) {
            float a = (float)c * w0;
            if (R >= 2) a = __builtin_fmaf((float)s2, w2, a);          // outermost pair first, as in the kernel
            a = __builtin_fmaf((float)s1, w1, a);
            double d = __dmul_rn((double)c, G.w[R]);
            if (R >= 2) d = __dadd_rn(d, __dmul_rn((double)s2, G.w[R - 2]));
            d = __dadd_rn(d, __dmul_rn((double)s1, G.w[R - 1]));
            const float err = __builtin_fabsf(a - (float)d);
            worst = __builtin_fmaxf(worst, err);
            nbad += err > e1;
        }
    // the horizontal stage, sampled: float32 intermediates v (the reference's), the kernel's inputs v~ = v +- e1 at worst,
    // the kernel's float32 sequence against the float64 sum over v: |a - A| <= (2 R + 4) u + 2^-15
    {
        const float bound = (float)(2 * R + 4) / 131072.0f + 1.0f / 32768.0f;
        unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * 256 + threadIdx.x + 1);
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
        for (int it = 0; it < 4096; it++) {
            float v[5], vt[5];
            const unsigned mode = rnd() & 3u;                       // smooth, rough, near-flat, extremes
            const float base = (float)(rnd() % 256u);
            for (int i = 0; i < 5; i++) {
                float x = mode == 0 ? base + (float)(rnd() % 1024u) / 64.0f - 8.0f : mode == 1 ? (float)(rnd() % 65536u) / 257.0f
                          : mode == 2 ? base + (float)(rnd() % 16u) / 65536.0f : ((rnd() & 1u) ? 255.0f : 0.0f);
                x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 255.0f);
                v[i] = x;
                const float d = (rnd() & 1u) ? e1 : -e1;
                vt[i] = __builtin_fminf(__builtin_fmaxf(x + ((rnd() & 3u) ? d : 0.0f), 0.0f), 255.0f);
            }
            float a = __builtin_fmaf(vt[2], w0, -0.5f);
            if (R >= 2) a = __builtin_fmaf(vt[0] + vt[4], w2, a);
            a = __builtin_fmaf(vt[1] + vt[3], w1, a);
            double A = __dmul_rn((double)v[2], G.w[R]);
            if (R >= 2) A = __dadd_rn(A, __dmul_rn(__dadd_rn((double)v[0], (double)v[4]), G.w[R - 2]));
            A = __dadd_rn(A, __dmul_rn(__dadd_rn((double)v[1], (double)v[3]), G.w[R - 1]));
            const float err = __builtin_fabsf((float)((double)a + 0.5 - A));
            nbad += err > bound;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    atomicMax(maxerr_bits, __float_as_uint(worst));
}

int gauss_fast_selftest(mrchip_ctx *ctx, hipStream_t s, const GaussW *d_w, unsigned long long *d_bad, unsigned *d_maxerr) {
    HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
    HIP_TRY(hipMemsetAsync(d_maxerr, 0, 4, s));
    hipLaunchKernelGGL(gauss_fast_selftest_kernel, dim3(256), dim3(256), 0, s, d_w, d_bad, d_maxerr);
    HIP_TRY(hipGetLastError());
    return 0;
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
// The float32 form's undecided band E(R) is derived for tables whose taps are non-negative and add up to 1: every
// partial sum then stays below 256 and half an ulp of it is 2^-17.  A caller-supplied table that is not like that (the
// entry points take any) must not go through it: wrong bytes would come out silently.
// float32 image in (pitch in floats), uint8 out = trunc(float32 result): the two streaming passes (radius 0: a copy + trunc)
int launch_gaussian_f32(mrchip_ctx *ctx, hipStream_t s, const float *src, int spitch, Plane dst, int w, int h,
                        const GaussW *d_weights, float *tmp, int tpitch) {
    dim3 grid(cdiv(w, 256), h, 1);
    LAUNCH(ctx, s, "gauss_v_f32", 8.0 * w * h,
           hipLaunchKernelGGL(gauss_v_f32_kernel, grid, dim3(256), 0, s, src, spitch, tmp, tpitch, w, h, d_weights));
    LAUNCH(ctx, s, "gauss_h", 5.0 * w * h,
           hipLaunchKernelGGL(gauss_h_kernel, grid, dim3(256), 0, s, tmp, tpitch, (size_t)0, dst.p, dst.pitch, dst.stride, w, h,
                              d_weights));
    return 0;
}

bool gauss_weights_allow_fast(const GaussW &g) {
    double sum = 0;
    for (int i = 0; i < 2 * g.radius + 1; i++) {
        if (!(g.w[i] >= 0.0)) return false;
        sum += g.w[i];
    }
    return sum >= 1.0 - 1e-9 && sum <= 1.0 + 1e-9;
}

int launch_gaussian_batch(mrchip_ctx *ctx, hipStream_t s, Plane src, Plane dst, int w, int h, const GaussW *d_weights,
                          float *tmp, int tpitch, size_t tstride, int npages, int max_radius, bool fast_ok) {
    if (max_radius >= 1 && max_radius <= GF_RMAX && w >= 2 * GF_RMAX && h >= 2 * GF_RMAX) {
        // d_weights must be padded to max_radius (gauss_pad_weights)
        dim3 gridf(cdiv(w, GF_TW), cdiv(h, GF_TH), npages);
        // the float32 form + its two clean-up launches, with the lists in the float scratch the two-pass kernels would
        // use (MRCHIP_GAUSS_FAST=0: the float64 tile kernel everywhere; the parity tests run both)
        const char *fast_env = getenv("MRCHIP_GAUSS_FAST");
        const size_t ntiles = (size_t)gridf.x * gridf.y * gridf.z;
        const size_t tmp_bytes = (npages > 1 ? tstride * (size_t)npages : (size_t)tpitch * h) * sizeof(float);
        const size_t need = ntiles * ((size_t)GQ_CAP + 1) * 4;
        if (fast_ok && !(fast_env && atoi(fast_env) == 0) && tmp && need <= tmp_bytes && ntiles < (1u << 31)) {
            unsigned *tcnt = reinterpret_cast<unsigned *>(tmp);                 // one count word per tile, written by every tile
            unsigned *tslots = tcnt + ntiles;                                   // GQ_CAP slots per tile
            const int cus = ctx->cus > 0 ? ctx->cus : 256;
            const int gx = (int)gridf.x, gy = (int)gridf.y;
#define GQ_CASE(RR)                                                                                              \
    case RR:                                                                                                     \
        LAUNCH(ctx, s, "gauss_fused", 2.0 * w * h * npages,                                                      \
               hipLaunchKernelGGL(gauss_fast_kernel<RR>, gridf, dim3(GF_THREADS), 0, s, src.p, src.pitch, src.stride, \
                                  dst.p, dst.pitch, dst.stride, w, h, d_weights, tcnt, tslots));                  \
        if (RR > GQ_INLINE_R)                                                                                     \
        LAUNCH(ctx, s, "gauss_fix", 0.0,                                                                          \
               hipLaunchKernelGGL(gauss_fix_kernel<RR>, dim3(std::min<size_t>((ntiles + 255) / 256, (size_t)cus * 8)), dim3(256), 0, s, src.p, src.pitch, src.stride, dst.p, \
                                  dst.pitch, dst.stride, w, h, d_weights, tcnt, tslots, (unsigned)ntiles, gx, gy)); \
        LAUNCH(ctx, s, "gauss_redo", 0.0,                                                                         \
               hipLaunchKernelGGL(gauss_exact_tiles_kernel<RR>, dim3((unsigned)std::min<size_t>((ntiles + GQ_REDO_CHUNK - 1) / GQ_REDO_CHUNK, (size_t)cus * 8)), dim3(GF_THREADS), 0, s, src.p, src.pitch, \
                                  src.stride, dst.p, dst.pitch, dst.stride, w, h, d_weights, tcnt, (unsigned)ntiles, gx, gy)); \
        break;
            switch (max_radius) { GQ_CASE(1) GQ_CASE(2) GQ_CASE(3) GQ_CASE(4) GQ_CASE(5) GQ_CASE(6) GQ_CASE(7) GQ_CASE(8) }
#undef GQ_CASE
            return 0;
        }
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
