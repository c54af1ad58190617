// Noise estimate (reference: mrc.py:52-55, 273-296 -> skimage.restoration.estimate_sigma
// -> pywt.dwtn(x, 'db2')['dd'] -> median(|dd[dd != 0]|) / ppf(0.75)).
//
// dwt_dd: one lane per detail coefficient.  The db2 high-pass analysis filter is
// applied along axis 0 and then axis 1 with PyWavelets' 'symmetric' extension and
// its exact accumulation order (sum starts at 0, products added one by one, no FMA;
// for outputs that overhang the right border the mirrored products come first with
// the filter index descending).  uint8 input is the float32 path (grayimgf,
// mrc.py:372); bool input is PyWavelets' float64 path (mrc.py:253-254).
//
// median: exact order statistics by MSB-first radix select on the IEEE bit patterns
// of |dd| (non-negative floats order like unsigned integers): 11-bit digit
// histograms in LDS flushed to global, one tiny scan kernel per digit.  Both middle
// ranks are tracked so that an even count averages the two middle values in the
// array's precision, as numpy.median does.
//
// Algorithmic bytes: 0.25*P for the page-level estimate (central half crop read once).
#include "mrchip_internal.h"

namespace mrchip {

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

// (global, not flat: the pointers of these kernels come out of job records, for which hipcc emits the slower flat_*
// instructions unless the address space is spelled out)
typedef unsigned unsigned_a1 __attribute__((aligned(1)));
typedef const unsigned_a1 __attribute__((address_space(1))) *gc_u32a1_p;
__device__ __forceinline__ unsigned ld_u32(const uint8_t *p) { return *(gc_u32a1_p)(uintptr_t)p; }

template <class T>
struct Db2 { T f[4]; };

// y[o] for i = 2*o + 1 over a length-N signal read through get(idx), idx in [0, N)
template <class T, class Get>
__device__ __forceinline__ T dwt_point(const Db2<T> &F, int N, int i, Get get) {
    constexpr int FL = 4;
    T sum = 0;
    int j = 0;
    if (i >= N) {
        // right extension first: x[N+t] = x[N-1-t]; filter index i-N-j descending
        while (i - j >= N) {
            int kk;
            for (kk = 0; kk < N && i - j >= N; j++, kk++) sum = add_rn(sum, mul_rn(F.f[i - N - j], get(N - 1 - kk)));
            for (kk = 0; kk < N && i - j >= N; j++, kk++) sum = add_rn(sum, mul_rn(F.f[i - N - j], get(kk)));
        }
    }
    for (; j <= i && j < FL; j++) sum = add_rn(sum, mul_rn(F.f[j], get(i - j)));
    while (j < FL) {   // left extension: x[-1-t] = x[t]
        int kk;
        for (kk = 0; kk < N && j < FL; j++, kk++) sum = add_rn(sum, mul_rn(F.f[j], get(kk)));
        for (kk = 0; kk < N && j < FL; j++, kk++) sum = add_rn(sum, mul_rn(F.f[j], get(N - 1 - kk)));
    }
    return sum;
}

constexpr int DIG = 11;
constexpr int NBIN = 1 << DIG;

struct SelState {
    unsigned long long prefix[2];
    unsigned long long rank[2];
    unsigned long long count;
    unsigned hist[2][NBIN];
    double sigma;
};

template <class T> struct Key;
template <> struct Key<float> {
    using U = unsigned;
    static constexpr int BITS = 32;
    __device__ static U of(float v) { return __float_as_uint(fabsf(v)); }
    __device__ static float val(U k) { return __uint_as_float(k); }
};
template <> struct Key<double> {
    using U = unsigned long long;
    static constexpr int BITS = 64;
    __device__ static U of(double v) { return (U)__double_as_longlong(fabs(v)); }
    __device__ static double val(U k) { return __longlong_as_double((long long)k); }
};

// Selection state is cleared by its own tiny launch: the transform below already adds the first
// digit's histogram into it.
__global__ __launch_bounds__(256) void sel_reset_kernel(const SigJob *jobs) {
    unsigned *z = reinterpret_cast<unsigned *>(jobs[blockIdx.x].scratch);
    for (int i = threadIdx.x; i < (int)(sizeof(SelState) / 4); i += 256) z[i] = 0;
}

// dd coefficients + the histogram of their first (most significant) radix digit.
// Lane = 4 adjacent coefficients of a row: their 4 x 10 input bytes come in as 3 (unaligned)
// dwords per row, the axis-0 sums of the 10 columns are shared by neighbouring outputs, and the
// four results leave together.  Same operation order per coefficient as the one-output form
// (axis 0: ((0 + f0*x[i]) + f1*x[i-1]) + ..., then axis 1 likewise); coefficients whose taps leave
// the array take PyWavelets' symmetric-extension order in dwt_point.  DWT_RPB rows per workgroup
// share one LDS histogram.
// F32SRC (round 6): the source is a float32 image (`pitch` still in bytes) -- mrc.estimate_noise on a float32 array that
// does not hold whole numbers 0..255 (mrc.py:273-296 takes any).  Every coefficient then goes through dwt_point, the
// one-output form in PyWavelets' own order; rare and small, so no fast path.
constexpr int DWT_RPB = 4;
template <class T, bool F32SRC = false>
__global__ __launch_bounds__(256) void dwt_dd_kernel(const SigJob *jobs, Db2<T> F, size_t dd_off) {
    const SigJob J = jobs[blockIdx.z];
    const int w = J.w, h = J.h, pitch = J.pitch;
    constexpr bool AS_BOOL = sizeof(T) == 8;               // the float64 path is PyWavelets' treatment of bool arrays
    const int w2 = (w + 3) / 2, h2 = (h + 3) / 2;
    __shared__ unsigned lh[NBIN];
    for (int i = threadIdx.x; i < NBIN; i += 256) lh[i] = 0;
    __syncthreads();
    const int m0 = (blockIdx.x * 256 + threadIdx.x) * 4;   // first of the lane's 4 columns of dd
    const uint8_t *src = J.src;
    T *dd = reinterpret_cast<T *>(J.scratch + dd_off);
    auto cv = [&](unsigned v) -> T { return AS_BOOL ? (T)(v ? 1 : 0) : (T)v; };
    auto px = [&](int yy, int xx) -> T {
        if constexpr (F32SRC) return (T) * reinterpret_cast<const float *>(src + (size_t)yy * pitch + (size_t)xx * 4);
        else return cv(src[(size_t)yy * pitch + xx]);
    };
    // The symmetric extension on the left / top is an index reflection with the taps still in
    // ascending filter order, and so is the right / bottom one as long as at most two taps overhang
    // (the two swapped products commute); three overhang only for the last coefficient of an odd
    // length, which keeps PyWavelets' order in dwt_point.  Tiny arrays go there entirely.
    const bool roomy = !F32SRC && w >= 8 && h >= 4;
    for (int rr = 0; rr < DWT_RPB; rr++) {
        const int k = blockIdx.y * DWT_RPB + rr;            // row of dd
        if (k >= h2 || m0 >= w2) break;
        const int iy = 2 * k + 1;
        const bool row_fast = roomy && iy <= h + 1;         // iy - j in [-2, h+1]: plain reflection
        T out[4];
        bool done[4] = {false, false, false, false};
        if (row_fast && 2 * m0 + 1 <= w + 1) {
            // bytes 2*m0-4 .. 2*m0+7 of rows iy-3 .. iy (reflected into the array); columns
            // 2*m0-2 .. 2*m0+7 are taps.  Bytes outside the row are loaded but not used.
            unsigned rw[4][3];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int yy = iy - j;
                yy = yy < 0 ? -1 - yy : (yy >= h ? 2 * h - 1 - yy : yy);
                const uint8_t *p = src + (size_t)yy * pitch + 2 * m0 - 4;
#pragma unroll
                for (int q = 0; q < 3; q++) rw[j][q] = ld_u32(p + 4 * q);
            }
            T t[12];
#pragma unroll
            for (int bi = 2; bi < 12; bi++) {
                T sum = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) sum = add_rn(sum, mul_rn(F.f[j], cv((rw[j][bi >> 2] >> (8 * (bi & 3))) & 0xffu)));
                t[bi] = sum;
            }
            if (m0 == 0) { t[2] = t[5]; t[3] = t[4]; }     // columns -2, -1 mirror columns 1, 0
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ix = 2 * (m0 + q) + 1;
                if (ix <= w + 1) {                          // taps ix-3 .. ix; columns w, w+1 mirror w-1, w-2
                    const T tv[4] = {ix == w + 1 ? t[2 * q + 2] : (ix == w ? t[2 * q + 4] : t[2 * q + 5]),
                                     ix == w + 1 ? t[2 * q + 3] : t[2 * q + 4], t[2 * q + 3], t[2 * q + 2]};
                    T sum = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) sum = add_rn(sum, mul_rn(F.f[j], tv[j]));
                    out[q] = sum;
                    done[q] = true;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!done[q] && m0 + q < w2) {
                const int ix = 2 * (m0 + q) + 1;
                auto tcol = [&](int xx) -> T { return dwt_point<T>(F, h, iy, [&](int yy) { return px(yy, xx); }); };
                out[q] = dwt_point<T>(F, w, ix, tcol);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (m0 + q < w2) {
                ((T __attribute__((address_space(1))) *)(uintptr_t)dd)[(size_t)k * w2 + m0 + q] = out[q];
                if (out[q] != (T)0) atomicAdd(&lh[(unsigned)(Key<T>::of(out[q]) >> (Key<T>::BITS - DIG))], 1u);
            }
    }
    __syncthreads();
    SelState *st = reinterpret_cast<SelState *>(J.scratch);
    for (int i = threadIdx.x; i < NBIN; i += 256) {
        const unsigned c = lh[i];
        if (c) { atomicAdd(&st->hist[0][i], c); atomicAdd(&st->hist[1][i], c); }
    }
}

// histogram of the digit at `shift` (width `bits`) among the non-zero |dd| whose higher
// bits equal prefix[r]
template <class T>
__global__ __launch_bounds__(256) void sel_hist_kernel(const SigJob *jobs, size_t dd_off, int shift, int bits, int first) {
    using U = typename Key<T>::U;
    const SigJob J = jobs[blockIdx.y];
    SelState *st = reinterpret_cast<SelState *>(J.scratch);
    const T *dd = reinterpret_cast<const T *>(J.scratch + dd_off);
    const size_t n = (size_t)((J.w + 3) / 2) * ((J.h + 3) / 2);
    __shared__ unsigned lh[2][NBIN];
    for (int i = threadIdx.x; i < 2 * NBIN; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    const U p0 = (U)st->prefix[0], p1 = (U)st->prefix[1];
    const unsigned mask = (1u << bits) - 1u;
    auto count = [&](T v) {
        if (v == (T)0) return;
        U key = Key<T>::of(v);
        unsigned d = (unsigned)(key >> shift) & mask;
        U hi = (shift + bits >= Key<T>::BITS) ? (U)0 : (key >> (shift + bits));
        if (first || hi == p0) atomicAdd(&lh[0][d], 1u);
        if (first || hi == p1) atomicAdd(&lh[1][d], 1u);
    };
    // 16 bytes per lane and load (the pass is a plain read of the coefficients: one scalar load per iteration left
    // every lane waiting for its own previous load, 2.4 TB/s), two loads in flight
    constexpr int V = 16 / sizeof(T);
    typedef T Vec __attribute__((ext_vector_type(V)));
    typedef const Vec __attribute__((address_space(1))) *gc_vec_p;
    const gc_vec_p dv = (gc_vec_p)(uintptr_t)dd;                  // dd starts 256-byte aligned in the scratch
    const size_t nv = n / V, stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < nv; i += 2 * stride) {
        const Vec a = dv[i], b = dv[i + stride];
#pragma unroll
        for (int k = 0; k < V; k++) count(a[k]);
#pragma unroll
        for (int k = 0; k < V; k++) count(b[k]);
    }
    for (; i < nv; i += stride) {
        const Vec a = dv[i];
#pragma unroll
        for (int k = 0; k < V; k++) count(a[k]);
    }
    for (size_t t = nv * V + (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) count(dd[t]);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NBIN; i += 256) {
        unsigned c = (&lh[0][0])[i];
        if (c) atomicAdd(&(&st->hist[0][0])[i], c);
    }
}

// one wave: locate the bins of both ranks, extend the prefixes, clear the histograms;
// on the last digit produce sigma.  Lane l owns bins [l*per, (l+1)*per).
template <class T>
__global__ __launch_bounds__(64) void sel_scan_kernel(const SigJob *jobs, int bits, int first, int last, double *d_sigma) {
    using U = typename Key<T>::U;
    SelState *st = reinterpret_cast<SelState *>(jobs[blockIdx.x].scratch);
    const int lane = threadIdx.x;
    const int nb = 1 << bits;
    const int per = (nb + 63) / 64;
    unsigned long long lsum[2] = {0, 0};
    for (int r = 0; r < 2; r++)
        for (int b = lane * per; b < min(nb, (lane + 1) * per); b++) lsum[r] += st->hist[r][b];
    unsigned long long count = st->count, rank[2] = {st->rank[0], st->rank[1]}, prefix[2] = {st->prefix[0], st->prefix[1]};
    if (first) {
        unsigned long long m = lsum[0];
        for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off);
        count = m;
        prefix[0] = prefix[1] = 0;
        if (m == 0) { rank[0] = rank[1] = 0; }
        else if (m & 1) { rank[0] = rank[1] = m / 2; }
        else { rank[0] = m / 2 - 1; rank[1] = m / 2; }
    }
    if (count) {
        for (int r = 0; r < 2; r++) {
            // exclusive prefix of the lane sums
            unsigned long long inc = lsum[r];
            for (int off = 1; off < 64; off <<= 1) {
                unsigned long long v = __shfl_up(inc, off);
                if (lane >= off) inc += v;
            }
            const unsigned long long exc = inc - lsum[r];
            const bool mine = rank[r] >= exc && rank[r] < inc;       // exactly one lane
            unsigned long long nr = 0, bsel = 0;
            if (mine) {
                unsigned long long cum = exc;
                int b = lane * per;
                for (; b < min(nb, (lane + 1) * per); b++) {
                    unsigned c = st->hist[r][b];
                    if (rank[r] < cum + c) break;
                    cum += c;
                }
                nr = rank[r] - cum;
                bsel = (unsigned long long)b;
            }
            const unsigned long long mask = __ballot(mine);
            const int src = __ffsll((long long)mask) - 1;
            nr = __shfl(nr, src);
            bsel = __shfl(bsel, src);
            rank[r] = nr;
            prefix[r] = (prefix[r] << bits) | bsel;
        }
    }
    __syncthreads();
    for (int i = lane; i < 2 * NBIN; i += 64) (&st->hist[0][0])[i] = 0;
    if (lane == 0) {
        st->count = count;
        st->rank[0] = rank[0]; st->rank[1] = rank[1];
        st->prefix[0] = prefix[0]; st->prefix[1] = prefix[1];
        if (last) {
            if (count == 0) {
                st->sigma = __longlong_as_double(0x7ff8000000000000LL);       // NaN: median of nothing
            } else {
                T a = Key<T>::val((U)prefix[0]), b = Key<T>::val((U)prefix[1]);
                T med = (count & 1) ? a : (T)(add_rn(a, b) / (T)2);           // numpy: mean of the two middle values
                st->sigma = (double)med / 0.6744897501960817;                 // scipy.stats.norm.ppf(0.75)
            }
            d_sigma[blockIdx.x] = st->sigma;
        }
    }
}

static size_t dd_offset() { return (sizeof(SelState) + 255) & ~(size_t)255; }

template <class T, bool F32SRC = false>
static int run_sigma(mrchip_ctx *ctx, hipStream_t s, const SigJob *h_jobs, const SigJob *d_jobs, int njobs,
                     double *d_sigma) {
    static const double HI[4] = {-0.48296291314453416, 0.8365163037378079, -0.2241438680420134,
                                 -0.12940952255126037};
    Db2<T> F;
    for (int i = 0; i < 4; i++) F.f[i] = (T)HI[i];
    int maxw2 = 0, maxh2 = 0;
    double alg = 0;
    size_t maxn = 0;
    for (int i = 0; i < njobs; i++) {
        const int w2 = (h_jobs[i].w + 3) / 2, h2 = (h_jobs[i].h + 3) / 2;
        maxw2 = std::max(maxw2, w2); maxh2 = std::max(maxh2, h2);
        maxn = std::max(maxn, (size_t)w2 * h2);
        alg += (double)h_jobs[i].w * h_jobs[i].h;
    }
    LAUNCH(ctx, s, "median_reset", 0.0, hipLaunchKernelGGL(sel_reset_kernel, dim3(njobs), dim3(256), 0, s, d_jobs));
    LAUNCH(ctx, s, sizeof(T) == 4 ? "dwt_dd_f32" : "dwt_dd_f64", alg,
           hipLaunchKernelGGL((dwt_dd_kernel<T, F32SRC>), dim3(cdiv(maxw2, 1024), cdiv(maxh2, DWT_RPB), njobs), dim3(256), 0, s,
                              d_jobs, F, dd_offset()));
    const int blocks = (int)std::min<size_t>(njobs > 8 ? 64 : 512, (maxn + 255) / 256);
    int shift = Key<T>::BITS;
    int first = 1;
    while (shift > 0) {
        const int bits = shift >= DIG ? DIG : shift;
        shift -= bits;
        const int last = shift == 0;
        if (!first)      // the first digit's histogram came with the transform
            LAUNCH(ctx, s, "median_hist", 0.0,
                   hipLaunchKernelGGL((sel_hist_kernel<T>), dim3(blocks, njobs), dim3(256), 0, s, d_jobs, dd_offset(), shift,
                                      bits, first));
        LAUNCH(ctx, s, "median_scan", 0.0,
               hipLaunchKernelGGL((sel_scan_kernel<T>), dim3(njobs), dim3(64), 0, s, d_jobs, bits, first, last, d_sigma));
        first = 0;
    }
    return 0;
}

size_t sigma_scratch_bytes(int w, int h, int kind) {
    const size_t n = (size_t)((w + 3) / 2) * ((h + 3) / 2);
    return ((dd_offset() + n * (kind == 1 ? sizeof(double) : sizeof(float)) + 255) & ~(size_t)255);
}

// h_jobs / d_jobs: the same njobs records on host and device; job i's result lands in d_sigma[i]
int launch_estimate_sigma_jobs(mrchip_ctx *ctx, hipStream_t s, const SigJob *h_jobs, const SigJob *d_jobs, int njobs,
                               int kind, double *d_sigma) {
    if (njobs <= 0) return 0;
    if (njobs > MAX_GRID_Z) {        // job index = grid.y / grid.z slice
        for (int o = 0; o < njobs; o += MAX_GRID_Z)
            TRY(launch_estimate_sigma_jobs(ctx, s, h_jobs + o, d_jobs + o, std::min(MAX_GRID_Z, njobs - o), kind, d_sigma + o));
        return 0;
    }
    for (int i = 0; i < njobs; i++)
        if (h_jobs[i].w <= 0 || h_jobs[i].h <= 0) { set_error("estimate_sigma: empty array"); return MRCHIP_E_ARG; }
    if (kind == 0) return run_sigma<float>(ctx, s, h_jobs, d_jobs, njobs, d_sigma);
    if (kind == 2) return run_sigma<float, true>(ctx, s, h_jobs, d_jobs, njobs, d_sigma);      // float32 source plane
    return run_sigma<double>(ctx, s, h_jobs, d_jobs, njobs, d_sigma);
}

}  // namespace mrchip
