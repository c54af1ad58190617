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
// v1 schedule: one workgroup per page-layer walks the rows; thread t owns the
// P=16 adjacent columns [16t, 16t+16) and keeps their vertical running sums
// (FIR: masked image sums + mask count over rows [ys,ye); IIR: output sums over
// rows [ys,y)) in registers.  Per row the threads publish their column sums
// (8 x u16 per column, one 16-byte LDS entry) and slide the horizontal windows
// over the LDS row.  All global loads a row needs are independent of the
// serial chain and are issued one row ahead.  int32 throughout, truncating
// division (exact: see div_small).  Batches run one workgroup per page-layer
// (256 CUs -> 128 pages' fg+bg concurrently).
//
// Algorithmic bytes: (1 + 2C)*w*h per call (mask + img in, out).
#include "mrchip_internal.h"

namespace mrchip {

constexpr int OP = 16;   // columns per thread

template <int C>
struct RowRegs {
    uint4 m;        // 16 mask bytes
    uint4 px[C];    // 16*C image bytes
};

template <int C>
__device__ __forceinline__ RowRegs<C> load_row_regs(const uint8_t *mask, int mpitch, const uint8_t *img, int ipitch,
                                                    int y, int x0, bool ok) {
    RowRegs<C> r;
    if (ok) {
        r.m = *reinterpret_cast<const uint4 *>(mask + (size_t)y * mpitch + x0);
        const uint4 *p = reinterpret_cast<const uint4 *>(img + (size_t)y * ipitch + (size_t)x0 * C);
#pragma unroll
        for (int i = 0; i < C; i++) r.px[i] = p[i];
    } else {
        r.m = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < C; i++) r.px[i] = make_uint4(0, 0, 0, 0);
    }
    return r;
}

__device__ __forceinline__ unsigned byte_of(const uint4 &v, int i) {
    unsigned d = (i >> 2) == 0 ? v.x : (i >> 2) == 1 ? v.y : (i >> 2) == 2 ? v.z : v.w;
    return (d >> (8 * (i & 3))) & 0xffu;
}

template <int C>
__device__ __forceinline__ unsigned px_byte(const uint4 (&px)[C], int j) {   // j-th byte of the 16*C block
    return byte_of(px[j >> 4], j & 15);
}

// val / cnt for 0 <= val <= 255*cnt, 1 <= cnt < 2^15: the quotient is <= 255 and its
// fractional part is a multiple of 1/cnt, so (val + 0.5) * rcp(cnt) truncates exactly
// (margin 0.5/cnt >> fp32 error of 255 * 2^-22).
__device__ __forceinline__ unsigned div_small(int val, float rc) {
    return (unsigned)(((float)val + 0.5f) * rc);
}

template <int C, int MAXT>
__global__ __launch_bounds__(MAXT) void optimise_kernel(const uint8_t *__restrict__ mask, int mpitch,
                                                        const uint8_t *__restrict__ img, int ipitch,
                                                        uint8_t *out, int opitch, int w, int h, int n, int inv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // entry for column c lives at index c + npad; npad leading / trailing entries stay zero
    constexpr int EW = (C == 3) ? 4 : 2;          // dwords per entry
    unsigned *ent = reinterpret_cast<unsigned *>(smem);
    const int npad = n;
    const int T = blockDim.x;
    const int t = threadIdx.x;
    const int x0 = t * OP;
    const int nent = T * OP + 2 * npad;
    for (int i = t; i < nent * EW; i += T) ent[i] = 0;
    __syncthreads();

    const bool act = x0 < w;                       // thread has at least one real column
    unsigned colok = 0;                            // bit i: column x0+i < w
#pragma unroll
    for (int i = 0; i < OP; i++) if (x0 + i < w) colok |= 1u << i;
    const unsigned invb = inv ? 1u : 0u;

    int fir[OP][C], firc[OP], iir[OP][C];
#pragma unroll
    for (int i = 0; i < OP; i++) {
        firc[i] = 0;
#pragma unroll
        for (int c = 0; c < C; c++) { fir[i][c] = 0; iir[i][c] = 0; }
    }
    uint4 prev[C];                                 // output row y-1 (16*C bytes)
#pragma unroll
    for (int c = 0; c < C; c++) prev[c] = make_uint4(0, 0, 0, 0);

    auto fir_apply = [&](const RowRegs<C> &r, int sign) {
#pragma unroll
        for (int i = 0; i < OP; i++) {
            unsigned mb = byte_of(r.m, i);
            bool on = (((mb != 0) ? 1u : 0u) ^ invb) && ((colok >> i) & 1u);
            if (on) {
                firc[i] += sign;
#pragma unroll
                for (int c = 0; c < C; c++) fir[i][c] += sign * (int)px_byte<C>(r.px, i * C + c);
            }
        }
    };

    // FIR rows [0, min(h, n-1)) enter before the loop; row y+n-1 enters at step y
    for (int yy = 0; yy < min(h, n - 1); yy++) {
        RowRegs<C> r = load_row_regs<C>(mask, mpitch, img, ipitch, yy, x0, act);
        fir_apply(r, +1);
    }
    // prefetch for y = 0
    RowRegs<C> r_enter = load_row_regs<C>(mask, mpitch, img, ipitch, n - 1, x0, act && (n - 1 < h) && n >= 1);
    RowRegs<C> r_leave = load_row_regs<C>(mask, mpitch, img, ipitch, 0, x0, false);
    RowRegs<C> r_cur = load_row_regs<C>(mask, mpitch, img, ipitch, 0, x0, act);
    uint4 o_leave[C];
#pragma unroll
    for (int c = 0; c < C; c++) o_leave[c] = make_uint4(0, 0, 0, 0);

    for (int y = 0; y < h; y++) {
        // ---- issue next row's loads first (independent of the serial chain) ----
        const int yn = y + 1;
        RowRegs<C> n_enter = load_row_regs<C>(mask, mpitch, img, ipitch, yn + n - 1, x0, act && (yn + n - 1 < h));
        RowRegs<C> n_leave = load_row_regs<C>(mask, mpitch, img, ipitch, yn - n - 1, x0, act && (yn - n - 1 >= 0));
        RowRegs<C> n_cur = load_row_regs<C>(mask, mpitch, img, ipitch, yn, x0, act && (yn < h));
        uint4 n_oleave[C];
        {
            const bool ok = act && (yn - n - 1 >= 0);
            const uint4 *p = reinterpret_cast<const uint4 *>(out + (size_t)max(yn - n - 1, 0) * opitch + (size_t)x0 * C);
#pragma unroll
            for (int c = 0; c < C; c++) n_oleave[c] = ok ? p[c] : make_uint4(0, 0, 0, 0);
        }

        // ---- vertical running sums for row y ----
        if (y + n - 1 < h && n >= 1) fir_apply(r_enter, +1);      // row y+n-1 enters (ye = min(h, y+n))
        if (y - n - 1 >= 0 && n >= 1) fir_apply(r_leave, -1);     // row y-n-1 leaves (ys = max(0, y-n))
        if (y >= 1 && n >= 1) {
#pragma unroll
            for (int i = 0; i < OP; i++)
                if ((colok >> i) & 1u) {
#pragma unroll
                    for (int c = 0; c < C; c++) iir[i][c] += (int)px_byte<C>(prev, i * C + c);
                }
        }
        if (y - n - 1 >= 0 && n >= 1) {
#pragma unroll
            for (int i = 0; i < OP; i++)
                if ((colok >> i) & 1u) {
#pragma unroll
                    for (int c = 0; c < C; c++) iir[i][c] -= (int)px_byte<C>(o_leave, i * C + c);
                }
        }
        const int ys = max(0, y - n);

        // ---- publish column sums ----
#pragma unroll
        for (int i = 0; i < OP; i++) {
            unsigned *e = ent + (size_t)(x0 + i + npad) * EW;
            if constexpr (C == 3) {
                uint4 v;
                v.x = (unsigned)fir[i][0] | ((unsigned)fir[i][1] << 16);
                v.y = (unsigned)fir[i][2] | ((unsigned)firc[i] << 16);
                v.z = (unsigned)iir[i][0] | ((unsigned)iir[i][1] << 16);
                v.w = (unsigned)iir[i][2];
                *reinterpret_cast<uint4 *>(e) = v;
            } else {
                uint2 v;
                v.x = (unsigned)fir[i][0] | ((unsigned)firc[i] << 16);
                v.y = (unsigned)iir[i][0];
                *reinterpret_cast<uint2 *>(e) = v;
            }
        }
        __syncthreads();

        // ---- horizontal sliding windows over the LDS row ----
        int fs[C], is[C], fc = 0;
#pragma unroll
        for (int c = 0; c < C; c++) { fs[c] = 0; is[c] = 0; }
        auto add_fir = [&](int col, int sign) {
            const unsigned *e = ent + (size_t)(col + npad) * EW;
            if constexpr (C == 3) {
                uint2 v = *reinterpret_cast<const uint2 *>(e);
                fs[0] += sign * (int)(v.x & 0xffffu); fs[1] += sign * (int)(v.x >> 16);
                fs[2] += sign * (int)(v.y & 0xffffu); fc += sign * (int)(v.y >> 16);
            } else {
                unsigned v = e[0];
                fs[0] += sign * (int)(v & 0xffffu); fc += sign * (int)(v >> 16);
            }
        };
        auto add_iir = [&](int col, int sign) {
            const unsigned *e = ent + (size_t)(col + npad) * EW;
            if constexpr (C == 3) {
                uint2 v = *reinterpret_cast<const uint2 *>(e + 2);
                is[0] += sign * (int)(v.x & 0xffffu); is[1] += sign * (int)(v.x >> 16);
                is[2] += sign * (int)(v.y & 0xffffu);
            } else {
                is[0] += sign * (int)e[1];
            }
        };
        // window of pixel x0: fir columns [x0-n, x0+n), iir columns [x0-n, x0)
        for (int j = -n; j < n; j++) add_fir(x0 + j, +1);
        for (int j = -n; j < 0; j++) add_iir(x0 + j, +1);

        uint4 res[C];
#pragma unroll
        for (int c = 0; c < C; c++) res[c] = r_cur.px[c];      // masked pixels keep the image value
        unsigned resb[OP * C];
#pragma unroll
        for (int i = 0; i < OP; i++) {
            const int x = x0 + i;
            const int xs = max(0, x - n);
            const int cnt = fc + (y - ys) * (x - xs);
            const unsigned mb = byte_of(r_cur.m, i);
            const bool masked = (((mb != 0) ? 1u : 0u) ^ invb) != 0;
            const float rc = __frcp_rn((float)max(cnt, 1));
#pragma unroll
            for (int c = 0; c < C; c++) {
                unsigned q = cnt > 0 ? div_small(fs[c] + is[c], rc) : 0u;
                resb[i * C + c] = masked ? px_byte<C>(r_cur.px, i * C + c) : q;
            }
            // slide to pixel x+1
            add_fir(x + n, +1);
            add_fir(x - n, -1);
            add_iir(x, +1);
            add_iir(x - n, -1);
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            unsigned d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int b = c * 16 + q * 4;
                d[q] = resb[b] | (resb[b + 1] << 8) | (resb[b + 2] << 16) | (resb[b + 3] << 24);
            }
            res[c] = make_uint4(d[0], d[1], d[2], d[3]);
        }
        // ---- store the row (full vectors inside the image, bytes at the right edge) ----
        if (act) {
            uint8_t *o = out + (size_t)y * opitch + (size_t)x0 * C;
            if (x0 + OP <= w) {
#pragma unroll
                for (int c = 0; c < C; c++) reinterpret_cast<uint4 *>(o)[c] = res[c];
            } else {
                const int nbytes = (w - x0) * C;
#pragma unroll
                for (int j = 0; j < OP * C; j++)
                    if (j < nbytes) o[j] = (uint8_t)resb[j];
            }
        }
#pragma unroll
        for (int c = 0; c < C; c++) prev[c] = res[c];
        __syncthreads();       // everyone is done reading the LDS row

        r_enter = n_enter; r_leave = n_leave; r_cur = n_cur;
#pragma unroll
        for (int c = 0; c < C; c++) o_leave[c] = n_oleave[c];
    }
}

int launch_optimise(mrchip_ctx *ctx, hipStream_t s, const uint8_t *mask, int mpitch, const uint8_t *img, int ipitch,
                    uint8_t *out, int opitch, int w, int h, int c, int n, int invert_mask) {
    if (c != 1 && c != 3) { set_error("optimise: channels must be 1 or 3"); return MRCHIP_E_ARG; }
    // n <= 32 keeps val < 2^23 and cnt <= 5120, the range in which div_small is exact
    if (n < 0 || n > 32) { set_error("optimise: n_size %d outside [0,32]", n); return MRCHIP_E_UNSUPPORTED; }
    if (w <= 0 || h <= 0) return 0;
    int T = round_up(cdiv(w, OP), 64);
    if (T > 1024) { set_error("optimise: width %d > %d not supported", w, 1024 * OP); return MRCHIP_E_UNSUPPORTED; }
    const int ew = (c == 3) ? 16 : 8;
    size_t lds = (size_t)(T * OP + 2 * n) * ew;
    if (lds > 160 * 1024) {
        set_error("optimise: width %d needs %zu bytes of LDS (> 160 KiB)", w, lds);
        return MRCHIP_E_UNSUPPORTED;
    }
    if ((mpitch & 15) || (ipitch & 15) || (opitch & 15)) { set_error("optimise: pitches must be multiples of 16"); return MRCHIP_E_ARG; }
    const double alg = (1.0 + 2.0 * c) * w * h;
#define OPT_LAUNCH(CC, MT, NAME)                                                                        \
    do {                                                                                                \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(optimise_kernel<CC, MT>),            \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        LAUNCH(ctx, s, NAME, alg,                                                                       \
               hipLaunchKernelGGL((optimise_kernel<CC, MT>), dim3(1), dim3(T), lds, s, mask, mpitch,    \
                                  img, ipitch, out, opitch, w, h, n, invert_mask));                     \
    } while (0)
    if (c == 3) {
        if (T <= 256) OPT_LAUNCH(3, 256, "optimise_rgb");
        else if (T <= 512) OPT_LAUNCH(3, 512, "optimise_rgb");
        else OPT_LAUNCH(3, 1024, "optimise_rgb");
    } else {
        if (T <= 256) OPT_LAUNCH(1, 256, "optimise_gray");
        else if (T <= 512) OPT_LAUNCH(1, 512, "optimise_gray");
        else OPT_LAUNCH(1, 1024, "optimise_gray");
    }
#undef OPT_LAUNCH
    return 0;
}

}  // namespace mrchip
