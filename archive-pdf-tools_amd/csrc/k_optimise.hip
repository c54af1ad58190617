// optimise_gray2 / optimise_rgb2 (reference: cython/optimiser.pyx:153-273, 280-429;
// spec optimise_gray / optimise_rgb pyx:22-146) for gfx950.
//
// For every pixel with mask==0, in raster order:
//   ys=max(0,y-n) ye=min(H,y+n) xs=max(0,x-n) xe=min(W,x+n)          (half open)
//   val = sum_{[ys,ye)x[xs,xe), mask!=0} img  +  sum_{[ys,y)x[xs,x)} out
//   cnt = #mask in that window + (y-ys)*(x-xs)
//   out = cnt>0 ? val/cnt : 0                                          (pyx:261-269)
// The second sum reads OUTPUT rows above the current one (never the current
// row), so rows are strictly sequential and all pixels of a row are parallel.
//
// Schedule: one workgroup per page-layer (blockIdx.x = job) walks the rows;
// thread t owns the P adjacent columns [P*t, P*t+P) and keeps their vertical
// running sums (FIR: masked image sums + mask count over rows [ys,ye); IIR:
// output sums over rows [ys,y)) in registers.  Per row the threads publish their
// column sums to two LDS rows and slide the horizontal windows over them.  The
// LDS rows carry one pad element per P columns so that the lane-strided accesses
// of a wave fall on distinct banks.  All global loads a row needs are
// independent of the serial chain and are issued one row ahead.  int32
// throughout, truncating division (exact: see div_small).  A batch is one launch:
// 2 jobs (fg, bg) per page, each on its own CU.
//
// Algorithmic bytes: (1 + 2C)*w*h per call (mask + img in, out).
#include "mrchip_internal.h"

namespace mrchip {

template <int C, int P>
struct RowRegs {
    unsigned m[P / 4];          // P mask bytes
    unsigned px[P * C / 4];     // P*C image bytes
};

template <int C, int P>
__device__ __forceinline__ RowRegs<C, P> load_row_regs(const uint8_t *mask, int mpitch, const uint8_t *img, int ipitch,
                                                       int y, int x0, bool ok) {
    RowRegs<C, P> r;
    if (ok) {
        const unsigned *pm = reinterpret_cast<const unsigned *>(mask + (size_t)y * mpitch + x0);
        const unsigned *pi = reinterpret_cast<const unsigned *>(img + (size_t)y * ipitch + (size_t)x0 * C);
#pragma unroll
        for (int i = 0; i < P / 4; i++) r.m[i] = pm[i];
#pragma unroll
        for (int i = 0; i < P * C / 4; i++) r.px[i] = pi[i];
    } else {
#pragma unroll
        for (int i = 0; i < P / 4; i++) r.m[i] = 0;
#pragma unroll
        for (int i = 0; i < P * C / 4; i++) r.px[i] = 0;
    }
    return r;
}

// j-th byte of a dword array (j is a compile-time constant after unrolling: never index
// register arrays dynamically -- hipcc 7.2 miscompiled the dynamic form at the right image edge)
template <int N>
__device__ __forceinline__ unsigned byte_at(const unsigned (&v)[N], int j) { return (v[j >> 2] >> (8 * (j & 3))) & 0xffu; }

// val / cnt for 0 <= val <= 255*cnt and cnt <= 5120 (n <= 32): the quotient is <= 255 and its
// fractional part is a multiple of 1/cnt, so (val + 0.5) * rcp(cnt) truncates to it exactly
// (val < 2^23 is exact in fp32; margin 0.5/cnt >= 9.7e-5 against an error below 255 * 2^-22 = 6.1e-5).
__device__ __forceinline__ unsigned div_small(int val, float rc) { return (unsigned)(((float)val + 0.5f) * rc); }

template <int C, int P, int MAXT>
__global__ __launch_bounds__(MAXT) void optimise_kernel(const OptJob *jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const OptJob J = jobs[blockIdx.x];
    const uint8_t *__restrict__ mask = J.mask;
    const uint8_t *__restrict__ img = J.img;
    uint8_t *out = J.out;
    const int mpitch = J.mpitch, ipitch = J.ipitch, opitch = J.opitch, w = J.w, h = J.h, n = J.n;
    const unsigned invb = J.invert ? 1u : 0u;

    constexpr int EW = (C == 3) ? 2 : 1;          // dwords per LDS element
    const int npad = n;
    const int T = blockDim.x;
    const int t = threadIdx.x;
    const int x0 = t * P;
    const int nent = T * P + 2 * npad;
    const int nelem = nent + nent / P + 1;
    unsigned *firA = reinterpret_cast<unsigned *>(smem);
    unsigned *iirA = firA + (size_t)nelem * EW;
    for (int i = t; i < 2 * nelem * EW; i += T) firA[i] = 0;
    __syncthreads();
    auto eidx = [&](int col) { const int e = col + npad; return (e + e / P) * EW; };

    const bool act = x0 < w;                       // thread has at least one real column
    unsigned colok = 0;                            // bit i: column x0+i < w
#pragma unroll
    for (int i = 0; i < P; i++) if (x0 + i < w) colok |= 1u << i;

    int fir[P][C], firc[P], iir[P][C];
#pragma unroll
    for (int i = 0; i < P; i++) {
        firc[i] = 0;
#pragma unroll
        for (int c = 0; c < C; c++) { fir[i][c] = 0; iir[i][c] = 0; }
    }
    unsigned prev[P * C / 4];                      // output row y-1
#pragma unroll
    for (int i = 0; i < P * C / 4; i++) prev[i] = 0;

    auto fir_apply = [&](const RowRegs<C, P> &r, int sign) {
#pragma unroll
        for (int i = 0; i < P; i++) {
            const unsigned mb = byte_at(r.m, i);
            const bool on = ((((mb != 0) ? 1u : 0u) ^ invb) != 0) && ((colok >> i) & 1u);
            if (on) {
                firc[i] += sign;
#pragma unroll
                for (int c = 0; c < C; c++) fir[i][c] += sign * (int)byte_at(r.px, i * C + c);
            }
        }
    };

    // FIR rows [0, min(h, n-1)) enter before the loop; row y+n-1 enters at step y
    for (int yy = 0; yy < min(h, n - 1); yy++) {
        RowRegs<C, P> r = load_row_regs<C, P>(mask, mpitch, img, ipitch, yy, x0, act);
        fir_apply(r, +1);
    }
    RowRegs<C, P> r_enter = load_row_regs<C, P>(mask, mpitch, img, ipitch, n - 1, x0, act && (n - 1 < h) && n >= 1);
    RowRegs<C, P> r_leave = load_row_regs<C, P>(mask, mpitch, img, ipitch, 0, x0, false);
    RowRegs<C, P> r_cur = load_row_regs<C, P>(mask, mpitch, img, ipitch, 0, x0, act);
    unsigned o_leave[P * C / 4];
#pragma unroll
    for (int i = 0; i < P * C / 4; i++) o_leave[i] = 0;

    for (int y = 0; y < h; y++) {
        // ---- issue next row's loads first (independent of the serial chain) ----
        const int yn = y + 1;
        RowRegs<C, P> n_enter = load_row_regs<C, P>(mask, mpitch, img, ipitch, yn + n - 1, x0, act && (yn + n - 1 < h));
        RowRegs<C, P> n_leave = load_row_regs<C, P>(mask, mpitch, img, ipitch, yn - n - 1, x0, act && (yn - n - 1 >= 0));
        RowRegs<C, P> n_cur = load_row_regs<C, P>(mask, mpitch, img, ipitch, yn, x0, act && (yn < h));
        unsigned n_oleave[P * C / 4];
        {
            const bool ok = act && (yn - n - 1 >= 0);
            const unsigned *p = reinterpret_cast<const unsigned *>(out + (size_t)max(yn - n - 1, 0) * opitch + (size_t)x0 * C);
#pragma unroll
            for (int i = 0; i < P * C / 4; i++) n_oleave[i] = ok ? p[i] : 0u;
        }

        // ---- vertical running sums for row y ----
        if (y + n - 1 < h && n >= 1) fir_apply(r_enter, +1);      // row y+n-1 enters (ye = min(h, y+n))
        if (y - n - 1 >= 0 && n >= 1) fir_apply(r_leave, -1);     // row y-n-1 leaves (ys = max(0, y-n))
        if (y >= 1 && n >= 1) {
#pragma unroll
            for (int i = 0; i < P; i++)
                if ((colok >> i) & 1u) {
#pragma unroll
                    for (int c = 0; c < C; c++) iir[i][c] += (int)byte_at(prev, i * C + c);
                }
        }
        if (y - n - 1 >= 0 && n >= 1) {
#pragma unroll
            for (int i = 0; i < P; i++)
                if ((colok >> i) & 1u) {
#pragma unroll
                    for (int c = 0; c < C; c++) iir[i][c] -= (int)byte_at(o_leave, i * C + c);
                }
        }
        const int ys = max(0, y - n);

        // ---- publish column sums ----
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int e = eidx(x0 + i);
            if constexpr (C == 3) {
                *reinterpret_cast<uint2 *>(firA + e) =
                    make_uint2((unsigned)fir[i][0] | ((unsigned)fir[i][1] << 16), (unsigned)fir[i][2] | ((unsigned)firc[i] << 16));
                *reinterpret_cast<uint2 *>(iirA + e) =
                    make_uint2((unsigned)iir[i][0] | ((unsigned)iir[i][1] << 16), (unsigned)iir[i][2]);
            } else {
                firA[e] = (unsigned)fir[i][0] | ((unsigned)firc[i] << 16);
                iirA[e] = (unsigned)iir[i][0];
            }
        }
        __syncthreads();

        // ---- horizontal sliding windows over the LDS rows ----
        int fs[C], is[C], fc = 0;
#pragma unroll
        for (int c = 0; c < C; c++) { fs[c] = 0; is[c] = 0; }
        auto add_fir = [&](int col, int sign) {
            const int e = eidx(col);
            if constexpr (C == 3) {
                uint2 v = *reinterpret_cast<const uint2 *>(firA + e);
                fs[0] += sign * (int)(v.x & 0xffffu); fs[1] += sign * (int)(v.x >> 16);
                fs[2] += sign * (int)(v.y & 0xffffu); fc += sign * (int)(v.y >> 16);
            } else {
                unsigned v = firA[e];
                fs[0] += sign * (int)(v & 0xffffu); fc += sign * (int)(v >> 16);
            }
        };
        auto add_iir = [&](int col, int sign) {
            const int e = eidx(col);
            if constexpr (C == 3) {
                uint2 v = *reinterpret_cast<const uint2 *>(iirA + e);
                is[0] += sign * (int)(v.x & 0xffffu); is[1] += sign * (int)(v.x >> 16);
                is[2] += sign * (int)(v.y & 0xffffu);
            } else {
                is[0] += sign * (int)iirA[e];
            }
        };
        // window of pixel x0: fir columns [x0-n, x0+n), iir columns [x0-n, x0)
        for (int j = -n; j < n; j++) add_fir(x0 + j, +1);
        for (int j = -n; j < 0; j++) add_iir(x0 + j, +1);

        unsigned resb[P * C];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int x = x0 + i;
            const int xs = max(0, x - n);
            const int cnt = fc + (y - ys) * (x - xs);
            const unsigned mb = byte_at(r_cur.m, i);
            const bool masked = (((mb != 0) ? 1u : 0u) ^ invb) != 0;
            const float rc = __frcp_rn((float)max(cnt, 1));
#pragma unroll
            for (int c = 0; c < C; c++) {
                unsigned q = cnt > 0 ? div_small(fs[c] + is[c], rc) : 0u;
                resb[i * C + c] = masked ? byte_at(r_cur.px, i * C + c) : q;   // masked pixels keep the image value
            }
            if (i + 1 < P) {      // slide to pixel x+1
                add_fir(x + n, +1);
                add_fir(x - n, -1);
                add_iir(x, +1);
                add_iir(x - n, -1);
            }
        }
        unsigned res[P * C / 4];
#pragma unroll
        for (int q = 0; q < P * C / 4; q++)
            res[q] = (resb[4 * q] & 0xffu) | ((resb[4 * q + 1] & 0xffu) << 8) | ((resb[4 * q + 2] & 0xffu) << 16) |
                     (resb[4 * q + 3] << 24);
        // ---- store the row (whole dwords inside the image, bytes at the right edge) ----
        if (act) {
            uint8_t *o = out + (size_t)y * opitch + (size_t)x0 * C;
            if (x0 + P <= w) {
#pragma unroll
                for (int q = 0; q < P * C / 4; q++) reinterpret_cast<unsigned *>(o)[q] = res[q];
            } else {
                const int nbytes = (w - x0) * C;
#pragma unroll
                for (int j = 0; j < P * C; j++)
                    if (j < nbytes) o[j] = (uint8_t)resb[j];
            }
        }
#pragma unroll
        for (int q = 0; q < P * C / 4; q++) prev[q] = res[q];
        __syncthreads();       // everyone is done reading the LDS rows

        r_enter = n_enter; r_leave = n_leave; r_cur = n_cur;
#pragma unroll
        for (int q = 0; q < P * C / 4; q++) o_leave[q] = n_oleave[q];
    }
}

struct OptGeom { int P, T; size_t lds; };

static int opt_geometry(int w, int c, int n, OptGeom *g) {
    // n <= 32 keeps val < 2^23 and cnt <= 5120, the range in which div_small is exact
    if (n < 0 || n > 32) { set_error("optimise: n_size %d outside [0,32]", n); return MRCHIP_E_UNSUPPORTED; }
    int P = 4;
    while (P < 16 && cdiv(w, P) > 1024) P *= 2;
    int T = round_up(cdiv(w, P), 64);
    if (T > 1024) { set_error("optimise: width %d > %d not supported", w, 1024 * 16); return MRCHIP_E_UNSUPPORTED; }
    const int ew = (c == 3) ? 16 : 8;             // FIR + IIR element bytes per column
    const int nent = T * P + 2 * n;
    size_t lds = (size_t)(nent + nent / P + 1) * ew;
    if (lds > 160 * 1024) { set_error("optimise: width %d needs %zu bytes of LDS (> 160 KiB)", w, lds); return MRCHIP_E_UNSUPPORTED; }
    g->P = P; g->T = T; g->lds = lds;
    return 0;
}

// d_jobs: njobs OptJob records in device memory, all with the same w, c (same geometry);
// n_max = the largest n_size among them (sizes the LDS rows)
int launch_optimise_jobs(mrchip_ctx *ctx, hipStream_t s, const OptJob *d_jobs, int njobs, int w, int h, int c, int n_max) {
    if (c != 1 && c != 3) { set_error("optimise: channels must be 1 or 3"); return MRCHIP_E_ARG; }
    if (w <= 0 || h <= 0 || njobs <= 0) return 0;
    OptGeom g;
    TRY(opt_geometry(w, c, n_max, &g));
    const double alg = (1.0 + 2.0 * c) * w * h * njobs;
#define OPT_LAUNCH(CC, PP, MT, NAME)                                                                     \
    do {                                                                                                \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(optimise_kernel<CC, PP, MT>),        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));           \
        LAUNCH(ctx, s, NAME, alg,                                                                       \
               hipLaunchKernelGGL((optimise_kernel<CC, PP, MT>), dim3(njobs), dim3(g.T), g.lds, s, d_jobs)); \
    } while (0)
#define OPT_PICK(CC, NAME)                                                          \
    do {                                                                            \
        if (g.P == 4) { if (g.T <= 512) OPT_LAUNCH(CC, 4, 512, NAME); else OPT_LAUNCH(CC, 4, 1024, NAME); }   \
        else if (g.P == 8) OPT_LAUNCH(CC, 8, 1024, NAME);                           \
        else OPT_LAUNCH(CC, 16, 1024, NAME);                                        \
    } while (0)
    if (c == 3) OPT_PICK(3, "optimise_rgb");
    else OPT_PICK(1, "optimise_gray");
#undef OPT_PICK
#undef OPT_LAUNCH
    return 0;
}

}  // namespace mrchip
