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

constexpr int GMAXR = 60;
struct GaussW { double w[2 * GMAXR + 1]; int radius; };

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i >= 0 && i < n) return i;
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

__global__ __launch_bounds__(256) void gauss_v_kernel(const uint8_t *src, int spitch, float *tmp, int tpitch,
                                                      int w, int h, GaussW G) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = G.radius;
    double acc = __dmul_rn((double)src[(size_t)y * spitch + x], G.w[r]);
    for (int j = -r; j < 0; j++) {
        double a = (double)src[(size_t)reflect_idx(y + j, h) * spitch + x];
        double b = (double)src[(size_t)reflect_idx(y - j, h) * spitch + x];
        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(a, b), G.w[r + j]));
    }
    tmp[(size_t)y * tpitch + x] = (float)acc;
}

__global__ __launch_bounds__(256) void gauss_h_kernel(const float *tmp, int tpitch, uint8_t *dst, int dpitch,
                                                      int w, int h, GaussW G) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = G.radius;
    const float *row = tmp + (size_t)y * tpitch;
    double acc = __dmul_rn((double)row[x], G.w[r]);
    for (int j = -r; j < 0; j++) {
        double a = (double)row[reflect_idx(x + j, w)];
        double b = (double)row[reflect_idx(x - j, w)];
        acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(a, b), G.w[r + j]));
    }
    dst[(size_t)y * dpitch + x] = (uint8_t)(float)acc;      // float32 result, astype(uint8) truncation
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

// tmp: h * tpitch floats of scratch
int launch_gaussian_u8_scratch(mrchip_ctx *ctx, hipStream_t s, const uint8_t *src, int spitch, uint8_t *dst, int dpitch,
                               int w, int h, const double *h_weights, int radius, float *tmp, int tpitch) {
    if (radius < 0 || radius > GMAXR) { set_error("gaussian: radius %d outside [0,%d]", radius, GMAXR); return MRCHIP_E_UNSUPPORTED; }
    if (radius == 0) {
        HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, w, h, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    GaussW G;
    memset(&G, 0, sizeof(G));
    G.radius = radius;
    for (int i = 0; i < 2 * radius + 1; i++) G.w[i] = h_weights[i];
    dim3 grid(cdiv(w, 256), h);
    LAUNCH(ctx, s, "gauss_v", 5.0 * w * h,
           hipLaunchKernelGGL(gauss_v_kernel, grid, dim3(256), 0, s, src, spitch, tmp, tpitch, w, h, G));
    LAUNCH(ctx, s, "gauss_h", 5.0 * w * h,
           hipLaunchKernelGGL(gauss_h_kernel, grid, dim3(256), 0, s, tmp, tpitch, dst, dpitch, w, h, G));
    return 0;
}

}  // namespace mrchip
