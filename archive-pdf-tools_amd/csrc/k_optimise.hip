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
#include <cstdlib>

#include <initializer_list>

#include "mrchip_internal.h"

namespace mrchip {

template <int C, int P>
struct RowRegs {
    unsigned m[P / 4];          // P mask bytes
    unsigned px[P * C / 4];     // P*C image bytes
};

typedef const unsigned __attribute__((address_space(1))) *gc_u32p;     // global (not flat) loads
typedef unsigned __attribute__((address_space(1))) *g_u32p;
typedef uint8_t __attribute__((address_space(1))) *g_u8p;              // global (not flat) byte stores: a flat access also counts on lgkmcnt

// Unconditional loads from a row clamped into the image: no exec-masked load blocks (hipcc puts an
// `s_waitcnt vmcnt(0)` behind every predicated load, one exposed memory round trip each).  Whether
// the row / the columns count is decided where the bytes are USED (wave-uniform row tests, column
// byte masks).  Columns >= w read the row's padding (or the next row), which the masks zero.
// vo_m / vo_px: the lane's byte offsets into the mask row and the pixel row.  Callers that keep them opaque to the
// optimiser inside their row loop (an empty asm) get `uniform row base + 32-bit lane offset` loads (SGPR-base
// addressing); otherwise LICM hoists `img + lane offset` as a 64-bit per-lane pointer and every row pays 64-bit
// vector multiply-adds for its addresses.
template <int C, int P>
__device__ __forceinline__ RowRegs<C, P> load_row_regs_at(const uint8_t *mask, int mpitch, const uint8_t *img, int ipitch,
                                                          int y, int h, unsigned vo_m, unsigned vo_px) {
    RowRegs<C, P> r;
    const int yc = min(max(y, 0), h - 1);
    gc_u32p pm = (gc_u32p)((mask + (size_t)yc * mpitch) + vo_m);
    gc_u32p pi = (gc_u32p)((img + (size_t)yc * ipitch) + vo_px);
#pragma unroll
    for (int i = 0; i < P / 4; i++) r.m[i] = pm[i];
#pragma unroll
    for (int i = 0; i < P * C / 4; i++) r.px[i] = pi[i];
    return r;
}
template <int C, int P>
__device__ __forceinline__ RowRegs<C, P> load_row_regs(const uint8_t *mask, int mpitch, const uint8_t *img, int ipitch,
                                                       int y, int h, int x0) {
    return load_row_regs_at<C, P>(mask, mpitch, img, ipitch, y, h, (unsigned)x0, (unsigned)(x0 * C));
}

// Same with the mask taken from 1-bpp rows: the lane's 4 columns are one nibble of a dword that 8 lanes share; the raw
// word is returned, the nibble is picked where the row is used
template <int C, int P>
__device__ __forceinline__ RowRegs<C, P> load_row_regs_bits(const unsigned *mbits, int mwpr, const uint8_t *img, int ipitch,
                                                            int y, int h, unsigned vo_m, unsigned vo_px) {
    static_assert(P == 4, "one nibble per lane");
    RowRegs<C, P> r;
    const int yc = min(max(y, 0), h - 1);
    gc_u32p pm = (gc_u32p)((const uint8_t *)(mbits + (size_t)yc * mwpr) + vo_m);
    gc_u32p pi = (gc_u32p)((img + (size_t)yc * ipitch) + vo_px);
    r.m[0] = pm[0];
#pragma unroll
    for (int i = 0; i < P * C / 4; i++) r.px[i] = pi[i];
    return r;
}

// j-th byte of a dword array (j is a compile-time constant after unrolling: never index
// register arrays dynamically -- hipcc 7.2 miscompiled the dynamic form at the right image edge)
template <int N>
__device__ __forceinline__ unsigned byte_at(const unsigned (&v)[N], int j) { return (v[j >> 2] >> (8 * (j & 3))) & 0xffu; }

// val / cnt for 0 <= val <= 255*cnt and cnt <= 5120 (n <= 32): the quotient is <= 255 and its
// fractional part is a multiple of 1/cnt, so (val + 0.5) * rcp(cnt) truncates to it exactly
// (val < 2^23 is exact in fp32; margin 0.5/cnt >= 9.7e-5 against an error below 255 * 1.5 * 2^-23 = 4.6e-5
// with the 1-ulp v_rcp_f32).
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
    // Threads past the image (the thread count is rounded up, P columns each) still issue their unconditional row
    // loads; they read the last column group that starts inside the image instead of P*C*(x0 - w) bytes past the row
    // -- which for the last rows of a small image lies beyond the allocation (fault found by tests/fuzz_parity.py).
    // Their bytes are masked out where they are used.
    const int xl = min(x0, max(0, ((w - 1) / P) * P));
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

    // byte masks of the valid columns: colm (one byte per column) and pxm (C bytes per column)
    unsigned colm[P / 4], pxm[P * C / 4];
#pragma unroll
    for (int q = 0; q < P / 4; q++) {
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) if ((colok >> (4 * q + b)) & 1u) m |= 0xffu << (8 * b);
        colm[q] = m;
    }
#pragma unroll
    for (int q = 0; q < P * C / 4; q++) {
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) if ((colok >> ((4 * q + b) / C)) & 1u) m |= 0xffu << (8 * b);
        pxm[q] = m;
    }
    // 0xFF per selected ("on") column: nonzero mask byte, optionally inverted, inside the image
    auto on_bytes = [&](const unsigned (&m)[P / 4], unsigned (&on)[P / 4]) {
#pragma unroll
        for (int q = 0; q < P / 4; q++) {
            unsigned t = (((m[q] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m[q]) & 0x80808080u;   // 0x80 per nonzero byte
            t = (t - (t >> 7)) | t;                                                       // -> 0xFF
            on[q] = (invb ? ~t : t) & colm[q];
        }
    };
    // expand one mask byte per column to C bytes per column (interleaved pixel layout)
    auto expand_px = [&](const unsigned (&on)[P / 4], unsigned (&e)[P * C / 4]) {
        if constexpr (C == 1) {
#pragma unroll
            for (int q = 0; q < P / 4; q++) e[q] = on[q];
        } else {
#pragma unroll
            for (int q = 0; q < P / 4; q++) {
                // bytes M0 M1 M2 M3 -> [M0 M0 M0 M1] [M1 M1 M2 M2] [M2 M3 M3 M3]
                e[3 * q + 0] = __builtin_amdgcn_perm(0u, on[q], 0x01000000u);
                e[3 * q + 1] = __builtin_amdgcn_perm(0u, on[q], 0x02020101u);
                e[3 * q + 2] = __builtin_amdgcn_perm(0u, on[q], 0x03030302u);
            }
        }
    };
    auto fir_apply = [&](const RowRegs<C, P> &r, int sign) {
        unsigned on[P / 4], e[P * C / 4];
        on_bytes(r.m, on);
        expand_px(on, e);
#pragma unroll
        for (int i = 0; i < P; i++) {
            firc[i] += sign * (int)(byte_at(on, i) & 1u);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int j = i * C + c;
                fir[i][c] += sign * (int)(((r.px[j >> 2] & e[j >> 2]) >> (8 * (j & 3))) & 0xffu);
            }
        }
    };

    // FIR rows [0, min(h, n-1)) enter before the loop; row y+n-1 enters at step y
    for (int yy = 0; yy < min(h, n - 1); yy++) {
        RowRegs<C, P> r = load_row_regs<C, P>(mask, mpitch, img, ipitch, yy, h, xl);
        fir_apply(r, +1);
    }
    RowRegs<C, P> r_enter = load_row_regs<C, P>(mask, mpitch, img, ipitch, n - 1, h, xl);
    RowRegs<C, P> r_leave = load_row_regs<C, P>(mask, mpitch, img, ipitch, 0, h, xl);
    RowRegs<C, P> r_cur = load_row_regs<C, P>(mask, mpitch, img, ipitch, 0, h, xl);
    unsigned o_leave[P * C / 4];
#pragma unroll
    for (int i = 0; i < P * C / 4; i++) o_leave[i] = 0;

    for (int y = 0; y < h; y++) {
        // ---- issue next row's loads first (independent of the serial chain) ----
        const int yn = y + 1;
        RowRegs<C, P> n_enter = load_row_regs<C, P>(mask, mpitch, img, ipitch, yn + n - 1, h, xl);
        RowRegs<C, P> n_leave = load_row_regs<C, P>(mask, mpitch, img, ipitch, yn - n - 1, h, xl);
        RowRegs<C, P> n_cur = load_row_regs<C, P>(mask, mpitch, img, ipitch, yn, h, xl);
        unsigned n_oleave[P * C / 4];
        {
            gc_u32p p = (gc_u32p)(out + (size_t)min(max(yn - n - 1, 0), h - 1) * opitch + (size_t)xl * C);
#pragma unroll
            for (int i = 0; i < P * C / 4; i++) n_oleave[i] = p[i];
        }

        // ---- vertical running sums for row y ----
        if (y + n - 1 < h && n >= 1) fir_apply(r_enter, +1);      // row y+n-1 enters (ye = min(h, y+n))
        if (y - n - 1 >= 0 && n >= 1) fir_apply(r_leave, -1);     // row y-n-1 leaves (ys = max(0, y-n))
        if (y >= 1 && n >= 1) {      // prev / o_leave carry zeros in the bytes of columns >= w
#pragma unroll
            for (int i = 0; i < P; i++)
#pragma unroll
                for (int c = 0; c < C; c++) iir[i][c] += (int)byte_at(prev, i * C + c);
        }
        if (y - n - 1 >= 0 && n >= 1) {
#pragma unroll
            for (int i = 0; i < P; i++)
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const int j = i * C + c;
                    iir[i][c] -= (int)(((o_leave[j >> 2] & pxm[j >> 2]) >> (8 * (j & 3))) & 0xffu);
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
        lds_barrier();

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

        unsigned qb[P * C];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int x = x0 + i;
            const int xs = max(0, x - n);
            const int cnt = fc + (y - ys) * (x - xs);
            const float rc = __builtin_amdgcn_rcpf((float)max(cnt, 1));   // 1 ulp: inside div_small's margin
#pragma unroll
            for (int c = 0; c < C; c++) qb[i * C + c] = cnt > 0 ? div_small(fs[c] + is[c], rc) : 0u;
            if (i + 1 < P) {      // slide to pixel x+1
                add_fir(x + n, +1);
                add_fir(x - n, -1);
                add_iir(x, +1);
                add_iir(x - n, -1);
            }
        }
        // masked pixels keep the image value (new_img = np.copy(img)), the others take the quotient
        unsigned res[P * C / 4];
        {
            unsigned on[P / 4], e[P * C / 4];
            on_bytes(r_cur.m, on);
            expand_px(on, e);
#pragma unroll
            for (int q = 0; q < P * C / 4; q++) {
                const unsigned qq = (qb[4 * q] & 0xffu) | ((qb[4 * q + 1] & 0xffu) << 8) | ((qb[4 * q + 2] & 0xffu) << 16) |
                                    (qb[4 * q + 3] << 24);
                res[q] = ((r_cur.px[q] & e[q]) | (qq & ~e[q])) & pxm[q];
            }
        }
        // ---- store the row (whole dwords inside the image, bytes at the right edge) ----
        if (act) {
            uint8_t *o = out + (size_t)y * opitch + (size_t)x0 * C;
            if (x0 + P <= w) {
#pragma unroll
                for (int q = 0; q < P * C / 4; q++) ((g_u32p)o)[q] = res[q];
            } else {
                const int nbytes = (w - x0) * C;
#pragma unroll
                for (int j = 0; j < P * C; j++)
                    if (j < nbytes) ((g_u8p)(uintptr_t)o)[j] = (uint8_t)(res[j >> 2] >> (8 * (j & 3)));
            }
        }
#pragma unroll
        for (int q = 0; q < P * C / 4; q++) prev[q] = res[q];
        lds_barrier();         // everyone is done reading the LDS rows

        r_enter = n_enter; r_leave = n_leave; r_cur = n_cur;
#pragma unroll
        for (int q = 0; q < P * C / 4; q++) o_leave[q] = n_oleave[q];
    }
}


// 16-bit halves added straight out of the packed pairs (SDWA operand selects): no shift / mask first
__device__ __forceinline__ unsigned add_w0w0(unsigned a, unsigned b) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned add_w1w1(unsigned a, unsigned b) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned add_dw0(unsigned a, unsigned b) {      // a + (b & 0xffff)
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned add_dw1(unsigned a, unsigned b) {      // a + (b >> 16)
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---------------------------------------------------------------------------------------------
// Packed variant (the one the reference's two call sites use: n=3 and n=10).
// Per column the running sums live in the registers in the very format of the LDS rows:
//   RGB : FIR entry {fr | fg<<16, fb | cnt<<16}, IIR entry {ir | ig<<16, ib}      (2 dwords each)
//   gray: FIR entry {f | cnt<<16},               IIR entry {i}                    (1 dword each)
// 16-bit lanes never overflow: a column sum is <= 2n*255, a horizontal window of NH halves is
// <= (2n/NH)*2n*255 <= 65535 for n <= 8 (NH=1) / n <= 11 (NH=2), the IIR window is <= n*n*255.
// So every vertical update, every horizontal slide is a plain 32-bit add/sub on packed pairs
// (lane-wise non-negative results: add first, subtract what was added before), pixel bytes are
// routed into the lanes with v_perm_b32, and the publish step is a straight ds_write_b64.
// NCT: n_size as a compile-time constant (the reference's 3 and 10: every window loop unrolls and
// every LDS access becomes `thread base + immediate offset`), or -1 for a run-time n.
// DB: the two LDS rows are double-buffered (row y publishes into buffer y&1), which removes the
// second barrier of a row; used when 2x the rows fit the 160 KiB of LDS.
// ---- column strips of one page-layer on different workgroups -------------------------------------------------
// The causal window of optimise looks LEFT and UP only: pixel (y, x) needs outputs of rows y-n .. y-1, columns
// x-n .. x-1, never of its own row.  So a row can be cut into strips, one workgroup each, as long as the strip to the
// right gets the n columns at its left boundary of every finished output row of its left-hand neighbour.  STRIP_HALO
// columns of halo on both sides: halo threads build the vertical FIR sums of their columns like everybody else
// (inputs only); the LEFT halo threads keep the vertical IIR sums of the neighbour's last columns, fed row by row
// through a mailbox in global memory instead of their own results.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): per (page-layer, boundary, row) three 16-byte granules
// {3 dwords of output bytes, tag}, written with ONE `sc0 sc1` (write-through) store each by the neighbour's last three
// threads and read with `sc0 sc1` loads: no fence, no separate flag, valid wherever the two workgroups run.  The tag is
// launch epoch << 16 | row, so nothing left in the buffer by an earlier launch can be mistaken for data.  The consumer
// asks for row y's granules one row early; in the steady state the neighbour is a row or two ahead and the data is
// there, otherwise it polls (bounded).  Left strips have the lower block index: they are dispatched first.
constexpr int STRIP_HALO = 12;         // >= n for the packed kernels (n <= 11), a multiple of 4
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // a register quad the inline asm can name
struct StripInfo {
    int S;                  // strips per page-layer (grid = jobs * S), 1 = whole rows
    int sw;                 // core columns per strip (multiple of 4)
    u32x4 *mail;            // [job][S-1][h][4] granules
    unsigned tagbase;       // epoch << 16
    unsigned *err;          // set when a poll gave up
};

__device__ __forceinline__ u32x4 mail_load(const u32x4 *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void mail_store(u32x4 *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// ---- bands of rows -------------------------------------------------------------------------------------------
// Only unselected pixels (mask == 0) are rewritten (pyx:195, 261-269); a row without one is a copy of the image row, and
// after n such rows in a row the state a later row depends on -- the vertical FIR sums over image rows [y-n, y+n) and
// the IIR sums over OUTPUT rows [y-n, y) -- is a function of the inputs alone (those output rows ARE image rows).  So
// maximal runs of rows that are separated by >= n rows without an unselected pixel ("bands") are independent jobs:
// a band [y0, y1) rebuilds its sums from image rows [y0-1-n, y0+n-1) and walks its rows exactly like a whole page;
// everything between bands is copied.  For the bg layer of a text page (selected = everything but the ink, n = 10)
// that is ~26 bands of ~47 rows instead of one chain of 3000; a layer without such gaps (fg: selected = the ink) is one
// band = the whole page.  A band job also copies the rows [c0, y0) in front of it and [y1, c1) behind it.
struct OptBand { int job, c0, y0, y1, c1, pad_[3]; };

template <int C, int NH, int NCT, bool DB, bool MB, bool STRIP = false>
__device__ __forceinline__ void optimise_packed_rows(const OptJob &J, unsigned char *smem, const StripInfo SI = StripInfo{1, 0, nullptr, 0, nullptr},
                                                     int job_index = 0, int strip = 0, const OptBand *band = nullptr, bool zero_lds = true) {
    constexpr int P = 4;
    constexpr int EW = (C == 3) ? 2 : 1;          // dwords per entry
    constexpr int ND = P * C / 4;                 // dwords of pixel bytes per thread-row
    // n <= 7: FIR + IIR sums of a column (and of a whole window) fit the 16-bit lanes together, so the second LDS row
    // carries FIR + IIR entries and ONE accumulator slides over the row: T(x+1) = T(x) + fir[x+n] + iir[x] - (fir+iir)[x-n]
    constexpr bool SUMROW = (NH == 1 && NCT >= 0 && NCT <= 7);
    // block sums in the pad slots: the window spans [x0-NCT, x0+NCT) = 2 + 4 + 4 | 4 + 4 + 2 columns for NCT = 10
    constexpr bool BLOCKSUM = (NH == 2 && NCT == 10);
    const uint8_t *__restrict__ mask = J.mask;
    const uint8_t *__restrict__ img = J.img;
    uint8_t *out = J.out;
    const int mpitch = J.mpitch, ipitch = J.ipitch, opitch = J.opitch, w = J.w, h = J.h;
    const int n = NCT >= 0 ? NCT : J.n;
    const unsigned invm = J.invert ? 0xffffffffu : 0u;
    const unsigned *mbits = J.mbits;
    const int mwpr = J.mwpr;
    unsigned vo_m = 0, vo_px = 0;          // lane offsets of the row loads, set below (kept opaque inside the row loop)
    auto load_row = [&](int yy, int) {
        if constexpr (MB) return load_row_regs_bits<C, P>(mbits, mwpr, img, ipitch, yy, h, vo_m, vo_px);
        else return load_row_regs_at<C, P>(mask, mpitch, img, ipitch, yy, h, vo_m, vo_px);
    };

    const int npad = n;
    const int T = blockDim.x;
    const int t = threadIdx.x;
    const int XS = STRIP ? strip * SI.sw : 0;                       // first core column of this workgroup
    const int XE = STRIP ? min(w, XS + SI.sw) : w;                  // one past its last core column
    const int x0 = STRIP ? XS - STRIP_HALO + t * P : t * P;         // absolute column of the thread (may be < 0 or >= w)
    const int wr = STRIP ? T * P : min(T * P, (w + 3) & ~3);        // columns that get an LDS entry (thread-relative)
    const int nent = wr + 2 * npad;
    const int nelem = nent + nent / P + 1;
    unsigned *firA0 = reinterpret_cast<unsigned *>(smem);
    unsigned *iirA0 = firA0 + (size_t)nelem * EW;
    // (a workgroup that walks several bands zeroes once: publishing never touches the pad entries left and right of the
    // image; the barrier also separates the previous band's last LDS reads from this band's first publish)
    if (zero_lds)
        for (int i = t; i < (DB ? 4 : 2) * nelem * EW; i += T) firA0[i] = 0;
    __syncthreads();
    unsigned *firA = firA0, *iirA = iirA0;
    const int yb0 = band ? band->y0 : 0, yb1 = band ? band->y1 : h;          // the rows this call computes
    // column c -> dword index of its entry (one pad entry per 4 columns: conflict-free lane stride).
    // With c = x0 + j and x0 = 4t: index = (5t + d + (d>>2)) * EW, d = j + n >= 0 -- `5t*EW` is the
    // thread's base, the rest folds to an immediate when n is a compile-time constant.
    const int ebase = 5 * t * EW;
    auto eidx = [&](int col) { const int d = col - x0 + npad; return ebase + (d + (d >> 2)) * EW; };
    // the pad slot inside the thread's own column group (after the own entry d with d % 4 == 3)
    constexpr int DPAD = NCT >= 0 ? NCT + ((3 - NCT) & 3) : 0;
    const int epad = ebase + (DPAD + (DPAD >> 2) + 1) * EW;

    const bool act = STRIP ? (x0 >= XS && x0 < XE) : (x0 < w);      // thread owns output columns
    const bool lhalo = STRIP && strip > 0 && x0 < XS && x0 >= XS - STRIP_HALO;     // carries the neighbour's IIR columns
    const bool producer = STRIP && strip + 1 < SI.S && x0 >= XS + SI.sw - STRIP_HALO && x0 < XS + SI.sw;
    const int xl = min(max(x0, 0), max(0, ((w - 1) / P) * P));      // column the unconditional row loads use
    vo_px = (unsigned)(xl * C);
    vo_m = MB ? (unsigned)((xl >> 5) * 4) : (unsigned)xl;
    unsigned colm = 0, pxm[ND];                   // 0xFF per valid column / per valid pixel byte
#pragma unroll
    for (int b = 0; b < 4; b++) if (x0 + b >= 0 && x0 + b < w) colm |= 0xffu << (8 * b);
#pragma unroll
    for (int q = 0; q < ND; q++) {
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { const int cc = x0 + (4 * q + b) / C; if (cc >= 0 && cc < w) m |= 0xffu << (8 * b); }
        pxm[q] = m;
    }
    // mailbox slots: what this thread writes (producer) / reads (left halo); other lanes point at a harmless slot
    u32x4 *mail_out = nullptr;
    const u32x4 *mail_in = nullptr;
    if constexpr (STRIP) {
        const size_t per_b = (size_t)h * 4;
        u32x4 *base = SI.mail + (size_t)job_index * (SI.S - 1) * per_b;
        const int lane_o = producer ? (x0 - (XS + SI.sw - STRIP_HALO)) / P : 3;
        const int lane_i = lhalo ? (x0 - (XS - STRIP_HALO)) / P : 3;
        mail_out = base + (size_t)min(strip, SI.S - 2) * per_b + lane_o;
        mail_in = base + (size_t)max(strip - 1, 0) * per_b + lane_i;
    }

    struct Ent { unsigned d[EW]; };
    auto eadd = [](Ent &a, const Ent &b) {
#pragma unroll
        for (int k = 0; k < EW; k++) a.d[k] += b.d[k];
    };
    auto esub = [](Ent &a, const Ent &b) {
#pragma unroll
        for (int k = 0; k < EW; k++) a.d[k] -= b.d[k];
    };
    auto lds_ld = [&](const unsigned *A, int col) {
        Ent e;
        const unsigned *p = A + eidx(col);
        if constexpr (EW == 2) { uint2 v = *reinterpret_cast<const uint2 *>(p); e.d[0] = v.x; e.d[1] = v.y; }
        else e.d[0] = p[0];
        return e;
    };

    // 0xFF per selected column (mask set, optional inversion, inside the image).  MB: `m` is the raw 1-bpp word, the
    // lane's 4 columns are one nibble of it: nibble -> (inversion, validity) -> one byte 0/1 per bit -> x 255
    const unsigned msh = (unsigned)xl & 31u;
    unsigned coln = 0;                             // bit i: column x0 + i is inside the image
#pragma unroll
    for (int b = 0; b < 4; b++) if (x0 + b >= 0 && x0 + b < w) coln |= 1u << b;
    const unsigned invn = J.invert ? 0xFu : 0u;
    auto on_bytes = [&](unsigned m) {
        if constexpr (MB) {
            const unsigned nib = ((m >> msh) ^ invn) & coln;
            return (__umul24(nib, 0x00204081u) & 0x01010101u) * 255u;       // (a full multiply: bit 24 is beyond __umul24)
        } else {
            unsigned tt = (((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m) & 0x80808080u;
            tt = (tt - (tt >> 7)) | tt;
            return (tt ^ invm) & colm;
        }
    };
    // FIR entries of the 4 columns of a row: masked pixel bytes + selection bit
    auto fir_entries = [&](const RowRegs<C, P> &r, const unsigned on, Ent (&e)[P]) {
        const unsigned on01 = on & 0x01010101u;
        if constexpr (C == 3) {
            const unsigned d0 = r.px[0] & __builtin_amdgcn_perm(0u, on, 0x01000000u);   // [M0 M0 M0 M1]
            const unsigned d1 = r.px[1] & __builtin_amdgcn_perm(0u, on, 0x02020101u);   // [M1 M1 M2 M2]
            const unsigned d2 = r.px[2] & __builtin_amdgcn_perm(0u, on, 0x03030302u);   // [M2 M3 M3 M3]
            e[0].d[0] = __builtin_amdgcn_perm(d1, d0, 0x0c010c00u); e[0].d[1] = __builtin_amdgcn_perm(on01, d0, 0x0c040c02u);
            e[1].d[0] = __builtin_amdgcn_perm(d1, d0, 0x0c040c03u); e[1].d[1] = __builtin_amdgcn_perm(on01, d1, 0x0c050c01u);
            e[2].d[0] = __builtin_amdgcn_perm(d2, d1, 0x0c030c02u); e[2].d[1] = __builtin_amdgcn_perm(on01, d2, 0x0c060c00u);
            e[3].d[0] = __builtin_amdgcn_perm(d2, d2, 0x0c020c01u); e[3].d[1] = __builtin_amdgcn_perm(on01, d2, 0x0c070c03u);
        } else {
            const unsigned d0 = r.px[0] & on;
            e[0].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c040c00u);
            e[1].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c050c01u);
            e[2].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c060c02u);
            e[3].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c070c03u);
        }
    };
    // IIR entries of the 4 columns of an output row (bytes of columns >= w must be zero)
    auto iir_entries = [&](const unsigned (&o)[ND], Ent (&e)[P]) {
        if constexpr (C == 3) {
            e[0].d[0] = __builtin_amdgcn_perm(o[1], o[0], 0x0c010c00u); e[0].d[1] = __builtin_amdgcn_perm(0u, o[0], 0x0c0c0c02u);
            e[1].d[0] = __builtin_amdgcn_perm(o[1], o[0], 0x0c040c03u); e[1].d[1] = __builtin_amdgcn_perm(0u, o[1], 0x0c0c0c01u);
            e[2].d[0] = __builtin_amdgcn_perm(o[2], o[1], 0x0c030c02u); e[2].d[1] = __builtin_amdgcn_perm(0u, o[2], 0x0c0c0c00u);
            e[3].d[0] = __builtin_amdgcn_perm(o[2], o[2], 0x0c020c01u); e[3].d[1] = __builtin_amdgcn_perm(0u, o[2], 0x0c0c0c03u);
        } else {
            e[0].d[0] = o[0] & 0xffu; e[1].d[0] = (o[0] >> 8) & 0xffu; e[2].d[0] = (o[0] >> 16) & 0xffu; e[3].d[0] = o[0] >> 24;
        }
    };

    Ent firE[P], iirE[P];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int k = 0; k < EW; k++) { firE[i].d[k] = 0; iirE[i].d[k] = 0; }
    unsigned prev[ND];
#pragma unroll
    for (int q = 0; q < ND; q++) prev[q] = 0;

    // A layer whose mask is sparse (fg: the ink, a few percent of the page) mostly sees rows in which none of the wave's
    // 256 columns is selected: such a row adds nothing to the FIR sums and is skipped for the whole wave
    const bool sparse = !J.invert;
    auto fir_apply = [&](const RowRegs<C, P> &r, bool plus) {
        const unsigned on = on_bytes(r.m[0]);
        if (sparse && !__any(on != 0u)) return;
        Ent e[P];
        fir_entries(r, on, e);
#pragma unroll
        for (int i = 0; i < P; i++) { if (plus) eadd(firE[i], e[i]); else esub(firE[i], e[i]); }
    };

    // rows in front of the band: copies of the image
    auto copy_rows = [&](int ya, int yb) {
        constexpr int CB = 8;                           // rows per batch; the next batch's loads go out before this one's stores
        auto ld = [&](int yy, unsigned (&v)[CB][ND]) {
#pragma unroll
            for (int k = 0; k < CB; k++) {
                gc_u32p p = (gc_u32p)((img + (size_t)min(yy + k, yb - 1) * ipitch) + vo_px);
#pragma unroll
                for (int q = 0; q < ND; q++) v[k][q] = p[q];
            }
        };
        auto st = [&](int yy, const unsigned (&v)[CB][ND]) {
            if (!act) return;
#pragma unroll
            for (int k = 0; k < CB; k++) {
                if (yy + k < yb) {             // (no early exit: the loop must unroll, v[k] are registers)
                    uint8_t *o = out + (size_t)(yy + k) * opitch + (size_t)x0 * C;
                    if (x0 + P <= XE) {
#pragma unroll
                        for (int q = 0; q < ND; q++) ((g_u32p)o)[q] = v[k][q];
                    } else {
                        const int nbytes = (XE - x0) * C;
#pragma unroll
                        for (int j = 0; j < P * C; j++)
                            if (j < nbytes) ((g_u8p)(uintptr_t)o)[j] = (uint8_t)(v[k][j >> 2] >> (8 * (j & 3)));
                    }
                }
            }
        };
        if (ya >= yb) return;
        unsigned va[CB][ND], vb[CB][ND];
        ld(ya, va);
        for (int yy = ya; yy < yb; yy += 2 * CB) {
            if (yy + CB < yb) ld(yy + CB, vb);
            st(yy, va);
            if (yy + CB < yb) {
                if (yy + 2 * CB < yb) ld(yy + 2 * CB, va);
                st(yy + CB, vb);
            }
        }
    };
    if (band && !J.skip_copy) copy_rows(band->c0, band->y0);
    if (yb1 > yb0) {
    // The sums in front of the band's first row yb0: FIR rows [yb0-1-n, yb0+n-1) (row y+n-1 enters, row y-n-1 leaves at
    // step y), IIR rows [yb0-n, yb0-1) -- image rows, the outputs there are copies -- and `prev` = row yb0-1, which step
    // yb0 adds.  A band at the top of the page starts from nothing, like the whole page.  Five rows in flight.
    {
        const int pa = max(0, yb0 - 1 - n), pb = min(h, yb0 + n - 1);
        constexpr int WB = 5;
        for (int yy = pa; yy < pb; yy += WB) {
            RowRegs<C, P> r[WB];
#pragma unroll
            for (int k = 0; k < WB; k++) r[k] = load_row(min(yy + k, pb - 1), xl);
#pragma unroll
            for (int k = 0; k < WB; k++) {
                const int yr = yy + k;
                if (yr < pb) fir_apply(r[k], true);
                if (yr < pb && yr >= yb0 - n && yr < yb0) {
                    unsigned o[ND];
#pragma unroll
                    for (int q = 0; q < ND; q++) o[q] = r[k].px[q] & pxm[q];
                    if (yr < yb0 - 1) {
                        Ent e[P];
                        iir_entries(o, e);
#pragma unroll
                        for (int i = 0; i < P; i++) eadd(iirE[i], e[i]);
                    } else {
#pragma unroll
                        for (int q = 0; q < ND; q++) prev[q] = o[q];
                    }
                }
            }
        }
    }
    // The inputs of a row (entering / leaving / current image row, leaving output row, STRIP: the neighbour's granules)
    // are loaded one row ahead.  Two sets of them alternate -- the row loop is unrolled by two, row y consumes one set
    // while its loads for row y+1 fill the other -- so no register is copied from a "next" to a "current" variable.
    struct RowSet {
        RowRegs<C, P> e, l, c;
        unsigned ol[ND];
        u32x4 mbp, mbl;            // STRIP: granules in flight
    };
    RowSet SA, SB;
    SA.e = load_row(yb0 + n - 1, xl);
    SA.l = load_row(yb0 - n - 1, xl);
    SA.c = load_row(yb0, xl);
    SA.mbp = u32x4{0, 0, 0, 0}; SA.mbl = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < ND; q++) SA.ol[q] = 0;

    // A row's result is stored at the TOP of the next iteration (it lives in `prev` anyway), behind a barrier that
    // makes the compiler wait for this row's prefetched loads first: vmcnt retires in order, so with the store issued
    // at the end of the row the wait for the next row's loads also waited for the store's write acknowledgement -- a
    // memory round trip in the serial chain of every row.
    auto store_row = [&](int yy, const unsigned (&res)[ND]) {
        if (!act) return;
        uint8_t *o = out + (size_t)yy * opitch + (size_t)x0 * C;
        if (x0 + P <= XE) {
            if constexpr (NCT >= 8 || NCT < 0) {
                // wide windows (the bg layer, n = 10): the row is re-read once, 11 rows later, by which time sixteen walkers per XCD
                // have pushed it out of the 4 MB L2 anyway -- stored non-temporally it does not evict the image rows on its way
#pragma unroll
                for (int q = 0; q < ND; q++) __builtin_nontemporal_store(res[q], (unsigned *)(uintptr_t)((g_u32p)o + q));
            } else {
#pragma unroll
                for (int q = 0; q < ND; q++) ((g_u32p)o)[q] = res[q];
            }
        } else {
            const int nbytes = (XE - x0) * C;
#pragma unroll
            for (int j = 0; j < P * C; j++)
                if (j < nbytes) ((g_u8p)(uintptr_t)o)[j] = (uint8_t)(res[j >> 2] >> (8 * (j & 3)));
        }
    };
    // count of causal output pixels in the window, (y - ys) * (x - xs), as a float per column: it only changes while
    // the window is still growing (rows 0..n), so the steady state finds it in registers
    float kf[P];
    int dxw[P];
#pragma unroll
    for (int i = 0; i < P; i++) { dxw[i] = (x0 + i) - max(0, x0 + i - n); kf[i] = (float)(min(yb0, n) * dxw[i]); }
    auto do_row = [&](const int y, RowSet &RS, RowSet &NS) {
        // ---- this row's inputs (loaded a row ago) have landed: only now the previous row's store and the next row's
        // loads go out ----
        {
#pragma unroll
            for (int q = 0; q < P / 4; q++) asm volatile("" : "+v"(RS.e.m[q]), "+v"(RS.l.m[q]), "+v"(RS.c.m[q]) : : "memory");
#pragma unroll
            for (int q = 0; q < ND; q++) asm volatile("" : "+v"(RS.e.px[q]), "+v"(RS.l.px[q]), "+v"(RS.c.px[q]), "+v"(RS.ol[q]) : : "memory");
        }
        if (y > yb0) store_row(y - 1, prev);         // (row yb0-1 in front of a band is a copy, written with the copies)
        // STRIP: the next row's granules.  The loads are asm statements the compiler takes for finished: they stay in
        // these locals, untouched, until the `s_waitcnt` at the bottom of this row, and only then move into the next
        // row's set (a register copy between an asynchronous load and its wait would read the old contents)
        u32x4 nmbp = {0, 0, 0, 0}, nmbl = {0, 0, 0, 0};
        if constexpr (STRIP) {
            if (t < 64 || producer) {              // wave 0 holds the left halo; the producers are the last three core threads
                const unsigned tag_prev = SI.tagbase + (unsigned)(y - 1);
                if (y >= 1 && producer)
                    mail_store(mail_out + (size_t)(y - 1) * 4, u32x4{prev[0], ND > 1 ? prev[ND > 1 ? 1 : 0] : 0u, ND > 2 ? prev[ND > 2 ? 2 : 0] : 0u, tag_prev});
                if (y >= 1 && strip > 0 && t < 64) {
                    // the neighbour's output row y-1 (asked for a row ago): poll until it is this launch's row y-1
                    int spins = 0;
                    while (__any(lhalo && RS.mbp.w != tag_prev)) {
                        __builtin_amdgcn_s_sleep(4);
                        RS.mbp = mail_load(mail_in + (size_t)(y - 1) * 4);
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(RS.mbp) : : "memory");
                        if (++spins > (1 << 21)) { if (t == 0) __hip_atomic_store(SI.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    }
                    if (lhalo) {
                        prev[0] = RS.mbp.x; if constexpr (ND > 1) prev[1] = RS.mbp.y; if constexpr (ND > 2) prev[2] = RS.mbp.z;
                        RS.ol[0] = RS.mbl.x; if constexpr (ND > 1) RS.ol[1] = RS.mbl.y; if constexpr (ND > 2) RS.ol[2] = RS.mbl.z;
                    }
                }
                if (strip > 0 && t < 64) {
                    nmbp = mail_load(mail_in + (size_t)min(y, h - 1) * 4);                       // prev of the next row
                    nmbl = mail_load(mail_in + (size_t)min(max(y - n, 0), h - 1) * 4);           // row (y+1)-n-1, seen before
                }
            }
        }
        const int yn = y + 1;
        asm volatile("" : "+v"(vo_m), "+v"(vo_px));          // see load_row_regs_at
        NS.e = load_row(yn + n - 1, xl);
        NS.l = load_row(yn - n - 1, xl);
        NS.c = load_row(yn, xl);
        {
            // the output row that leaves the IIR sums at the next step; in front of the band it is the image row
            const int yl = min(max(yn - n - 1, 0), h - 1);
            const uint8_t *lrow = yl < yb0 ? img + (size_t)yl * ipitch : out + (size_t)yl * opitch;
            gc_u32p p = (gc_u32p)(lrow + vo_px);
#pragma unroll
            for (int q = 0; q < ND; q++) NS.ol[q] = p[q];
        }

        // ---- vertical running sums for row y (wave-uniform row tests) ----
        if (y + n - 1 < h && n >= 1) fir_apply(RS.e, true);        // ye = min(h, y+n)
        if (y - n - 1 >= 0 && n >= 1) fir_apply(RS.l, false);      // ys = max(0, y-n)
        if (y >= 1 && n >= 1) {
            Ent e[P];
            iir_entries(prev, e);
#pragma unroll
            for (int i = 0; i < P; i++) eadd(iirE[i], e[i]);
        }
        if (y - n - 1 >= 0 && n >= 1 && y > yb0) {          // (a band's first row: its IIR sums were built without that row)
            unsigned ol[ND];
#pragma unroll
            for (int q = 0; q < ND; q++) ol[q] = RS.ol[q] & pxm[q];
            Ent e[P];
            iir_entries(ol, e);
#pragma unroll
            for (int i = 0; i < P; i++) esub(iirE[i], e[i]);
        }
        if (y <= n) {
            asm volatile("" ::: "memory");        // a real (wave-uniform) branch: keeps the multiplies out of the steady state
#pragma unroll
            for (int i = 0; i < P; i++) kf[i] = (float)(y * dxw[i]);          // y - ys = y while y <= n
        }

        // ---- publish: the registers already hold the LDS entry format ----
        if constexpr (DB) {
            firA = firA0 + (size_t)(y & 1) * 2 * nelem * EW;
            iirA = firA + (size_t)nelem * EW;
        }
        if (STRIP || x0 < wr)
#pragma unroll
        for (int i = 0; i < P; i++) {
            const int e = eidx(x0 + i);
            Ent second = iirE[i];
            if constexpr (SUMROW) eadd(second, firE[i]);
            if constexpr (EW == 2) {
                *reinterpret_cast<uint2 *>(firA + e) = make_uint2(firE[i].d[0], firE[i].d[1]);
                *reinterpret_cast<uint2 *>(iirA + e) = make_uint2(second.d[0], second.d[1]);
            } else {
                firA[e] = firE[i].d[0];
                iirA[e] = second.d[0];
            }
        }
        // n = 10 (two half-window accumulators): each thread also leaves the SUM of its four entries in the pad slot that lies
        // inside its column group, so that a window starts from 4 block sums + 4 entries instead of 30 entries
        // (4 columns x 20 rows x 255 = 20 400 per 16-bit lane)
        Ent blkF, blkI;
        if constexpr (BLOCKSUM) {
            blkF = firE[0]; eadd(blkF, firE[1]); eadd(blkF, firE[2]); eadd(blkF, firE[3]);
            blkI = iirE[0]; eadd(blkI, iirE[1]); eadd(blkI, iirE[2]); eadd(blkI, iirE[3]);
            if (STRIP || x0 < wr) {
                if constexpr (EW == 2) {
                    *reinterpret_cast<uint2 *>(firA + epad) = make_uint2(blkF.d[0], blkF.d[1]);
                    *reinterpret_cast<uint2 *>(iirA + epad) = make_uint2(blkI.d[0], blkI.d[1]);
                } else {
                    firA[epad] = blkF.d[0];
                    iirA[epad] = blkI.d[0];
                }
            }
        }
        lds_barrier();

        // Only pixels with mask==0 get a quotient.  When every pixel of this wave's 256 columns is masked
        // in this row (most of the bg layer: the inverted mask is set wherever there is no ink) the
        // horizontal windows and the divisions are skipped for the whole wave; the row is a copy.
        const unsigned on_cur = on_bytes(RS.c.m[0]);
        unsigned qd[ND];
#pragma unroll
        for (int q = 0; q < ND; q++) qd[q] = 0;
        if (__any(act && on_cur != colm)) {
        // ---- horizontal windows of pixel x0: FIR [x0-n, x0+n) as NH packed halves, IIR [x0-n, x0) ----
        Ent aL, aR, aI;
#pragma unroll
        for (int k = 0; k < EW; k++) { aL.d[k] = 0; aR.d[k] = 0; aI.d[k] = 0; }
        if constexpr (SUMROW) {
#pragma unroll
            for (int j = -NCT; j < 0; j++) eadd(aL, lds_ld(iirA, x0 + j));          // (fir + iir)[x0 + j]
#pragma unroll
            for (int j = 0; j < NCT; j++) eadd(aL, j < P ? firE[j < P ? j : 0] : lds_ld(firA, x0 + j));
        } else if constexpr (BLOCKSUM) {
            // columns [x0-10, x0): two entries + the block sums of the two threads to the left; [x0, x0+10): the own block
            // (registers), the right-hand neighbour's, two entries
            auto pad_ld = [&](const unsigned *A, int k) {          // block sum of thread t + k
                Ent e;
                const unsigned *p = A + epad + 5 * k * EW;
                if constexpr (EW == 2) { uint2 v = *reinterpret_cast<const uint2 *>(p); e.d[0] = v.x; e.d[1] = v.y; }
                else e.d[0] = p[0];
                return e;
            };
            aL = lds_ld(firA, x0 - NCT); eadd(aL, lds_ld(firA, x0 - NCT + 1)); eadd(aL, pad_ld(firA, -2)); eadd(aL, pad_ld(firA, -1));
            aI = lds_ld(iirA, x0 - NCT); eadd(aI, lds_ld(iirA, x0 - NCT + 1)); eadd(aI, pad_ld(iirA, -2)); eadd(aI, pad_ld(iirA, -1));
            aR = blkF; eadd(aR, pad_ld(firA, 1)); eadd(aR, lds_ld(firA, x0 + NCT - 2)); eadd(aR, lds_ld(firA, x0 + NCT - 1));
        } else if constexpr (NCT >= 0) {
#pragma unroll
            for (int j = -NCT; j < 0; j++) { eadd(aL, lds_ld(firA, x0 + j)); eadd(aI, lds_ld(iirA, x0 + j)); }
#pragma unroll
            for (int j = 0; j < NCT; j++) {
                // own columns come from registers
                const Ent e = j < P ? firE[j < P ? j : 0] : lds_ld(firA, x0 + j);
                if constexpr (NH == 2) eadd(aR, e); else eadd(aL, e);
            }
        } else {
            for (int j = -n; j < 0; j++) { eadd(aL, lds_ld(firA, x0 + j)); eadd(aI, lds_ld(iirA, x0 + j)); }
            for (int j = 0; j < n; j++) {
                if constexpr (NH == 2) eadd(aR, lds_ld(firA, x0 + j)); else eadd(aL, lds_ld(firA, x0 + j));
            }
        }

#pragma unroll
        for (int i = 0; i < P; i++) {
            const int x = x0 + i;
            float fsum[C], fcnt;
            if constexpr (NH == 1 && NCT >= 0 && NCT <= 7) {
                // FIR + IIR window sums stay below 2^16 for n <= 7 (14*14*255 + 49*255): add the packed
                // pairs first, then split (the conversions take the 16-bit halves directly)
                const Ent T = aL;          // SUMROW: the one accumulator
                if constexpr (C == 3) {
                    fsum[0] = (float)(T.d[0] & 0xffffu); fsum[1] = (float)(T.d[0] >> 16);
                    fsum[2] = (float)(T.d[1] & 0xffffu); fcnt = (float)(T.d[1] >> 16);     // the IIR entry has no count half
                } else {
                    // gray: FIR {f | cnt<<16}, IIR {i} (a full dword, may exceed 16 bits only for n > 7)
                    fsum[0] = (float)(T.d[0] & 0xffffu); fcnt = (float)(T.d[0] >> 16);
                }
            } else if constexpr (C == 3 && NH == 2) {
                // three packed pairs per sum (left half, right half, IIR): the halves are added as they are
                fsum[0] = (float)add_dw0(add_w0w0(aL.d[0], aR.d[0]), aI.d[0]);
                fsum[1] = (float)add_dw1(add_w1w1(aL.d[0], aR.d[0]), aI.d[0]);
                fsum[2] = (float)add_dw0(add_w0w0(aL.d[1], aR.d[1]), aI.d[1]);
                fcnt = (float)add_w1w1(aL.d[1], aR.d[1]);
            } else if constexpr (C == 3) {
                fsum[0] = (float)((aL.d[0] & 0xffffu) + (aI.d[0] & 0xffffu));
                fsum[1] = (float)((aL.d[0] >> 16) + (aI.d[0] >> 16));
                fsum[2] = (float)((aL.d[1] & 0xffffu) + (aI.d[1] & 0xffffu));
                fcnt = (float)(aL.d[1] >> 16);
            } else {
                unsigned f0 = (aL.d[0] & 0xffffu) + aI.d[0], c0 = aL.d[0] >> 16;
                if constexpr (NH == 2) { f0 += aR.d[0] & 0xffffu; c0 += aR.d[0] >> 16; }
                fsum[0] = (float)f0; fcnt = (float)c0;
            }
            // cnt = fcnt + (y - ys) * (x - xs), all small integers: exact in fp32
            const float rc = __builtin_amdgcn_rcpf(__builtin_fmaxf(fcnt + kf[i], 1.0f));   // 1 ulp: inside the margin (div_small)
            // floor(v / cnt) = round-to-nearest-even((v + 0.5) / cnt - 0.5): the argument is at least 0.5/cnt away from
            // every half-integer, the arithmetic error stays below that (exhaustive device self-test), and
            // v_cvt_pk_u8_f32 rounds to nearest even and drops the byte into place (tools/ubench/cvt_pk_u8.hip).
            // cnt == 0 implies v == 0: rc = 1, offset 0, quotient 0 = the reference's `else: 0` (pyx:266-269)
            const float qoff = __builtin_fmaf(rc, 0.5f, -0.5f);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const int jb = i * C + c;
                qd[jb >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(fsum[c], rc, qoff), jb & 3, qd[jb >> 2]);
            }
            if (i + 1 < P) {      // slide to pixel x+1 (own columns' entries come from registers)
                if constexpr (SUMROW) {
                    eadd(aL, (i + NCT < P) ? firE[(i + NCT < P) ? i + NCT : 0] : lds_ld(firA, x + n));
                    eadd(aL, iirE[i]);
                    esub(aL, lds_ld(iirA, x - n));
                } else {
                    if constexpr (NH == 2) {
                        eadd(aL, firE[i]); esub(aL, lds_ld(firA, x - n));
                        eadd(aR, lds_ld(firA, x + n)); esub(aR, firE[i]);
                    } else {
                        eadd(aL, lds_ld(firA, x + n)); esub(aL, lds_ld(firA, x - n));
                    }
                    eadd(aI, iirE[i]); esub(aI, lds_ld(iirA, x - n));
                }
            }
        }
        }
        // masked pixels keep the image value (new_img = np.copy(img)), the others take the quotient
        unsigned res[ND];
        {
            const unsigned on = on_cur;
            if constexpr (C == 3) {
                const unsigned e0 = __builtin_amdgcn_perm(0u, on, 0x01000000u), e1 = __builtin_amdgcn_perm(0u, on, 0x02020101u),
                               e2 = __builtin_amdgcn_perm(0u, on, 0x03030302u);
                res[0] = ((RS.c.px[0] & e0) | (qd[0] & ~e0)) & pxm[0];
                res[1] = ((RS.c.px[1] & e1) | (qd[1] & ~e1)) & pxm[1];
                res[2] = ((RS.c.px[2] & e2) | (qd[2] & ~e2)) & pxm[2];
            } else {
                res[0] = ((RS.c.px[0] & on) | (qd[0] & ~on)) & pxm[0];
            }
        }
#pragma unroll
        for (int q = 0; q < ND; q++) prev[q] = res[q];
        if constexpr (!DB) lds_barrier();   // everyone is done reading the LDS rows (DB: the next row uses the other buffer)

        if constexpr (STRIP) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(nmbp), "+v"(nmbl) : : "memory");
            NS.mbp = nmbp; NS.mbl = nmbl;
        }
    };
    for (int y = yb0; y < yb1; y += 2) {
        do_row(y, SA, SB);
        if (y + 1 < yb1) do_row(y + 1, SB, SA);
    }
    store_row(yb1 - 1, prev);
    }
    if (band && !J.skip_copy) copy_rows(band->y1, band->c1);
}

// Rows of 4097..8160 columns: the same packed scheme with TWO groups of 4 columns per thread (columns
// 4t.. and 4(t+T)..), so that 1024 threads still cover the row and the LDS rows (one entry per column) stay
// within 160 KiB.  The per-group state doubles, so the next row's loads are not held in registers a row
// ahead (the 128-VGPR budget of a 1024-thread workgroup): each row loads what it needs for both groups up
// front.  Same arithmetic, same LDS layout (entry index = column + column/4), two barriers per row.
template <int C, int NH, int NCT>
__device__ __forceinline__ void optimise_packed_wide_rows(const OptJob &J, unsigned char *smem) {
    constexpr int P = 4, G = 2;
    constexpr int EW = (C == 3) ? 2 : 1;
    constexpr int ND = P * C / 4;
    const uint8_t *__restrict__ mask = J.mask;
    const uint8_t *__restrict__ img = J.img;
    uint8_t *out = J.out;
    const int mpitch = J.mpitch, ipitch = J.ipitch, opitch = J.opitch, w = J.w, h = J.h;
    const int n = NCT >= 0 ? NCT : J.n;
    const unsigned invm = J.invert ? 0xffffffffu : 0u;
    const int npad = n;
    const int T = blockDim.x, t = threadIdx.x;
    const int wr = min(G * T * P, (w + 3) & ~3);
    const int nent = wr + 2 * npad;
    const int nelem = nent + nent / P + 1;
    unsigned *firA = reinterpret_cast<unsigned *>(smem);
    unsigned *iirA = firA + (size_t)nelem * EW;
    for (int i = t; i < 2 * nelem * EW; i += T) firA[i] = 0;
    __syncthreads();
    struct Ent { unsigned d[EW]; };
    auto eadd = [](Ent &a, const Ent &b) {
#pragma unroll
        for (int k = 0; k < EW; k++) a.d[k] += b.d[k];
    };
    auto esub = [](Ent &a, const Ent &b) {
#pragma unroll
        for (int k = 0; k < EW; k++) a.d[k] -= b.d[k];
    };
    // column -> dword index of its entry (one pad entry per 4 columns)
    auto eidx = [&](int col) { const int d = col + npad; return (d + (d >> 2)) * EW; };
    auto lds_ld = [&](const unsigned *A, int col) {
        Ent e;
        const unsigned *p = A + eidx(col);
        if constexpr (EW == 2) { uint2 v = *reinterpret_cast<const uint2 *>(p); e.d[0] = v.x; e.d[1] = v.y; }
        else e.d[0] = p[0];
        return e;
    };
    int x0[G], xl[G];          // xl: column the unconditional row loads use (clamped into the image, see optimise_kernel)
    unsigned colm[G], pxm[G][ND];
    Ent firE[G][P], iirE[G][P];
    unsigned prev[G][ND];
#pragma unroll
    for (int g = 0; g < G; g++) {
        x0[g] = (t + g * T) * P;
        xl[g] = min(x0[g], max(0, ((w - 1) / P) * P));
        colm[g] = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) if (x0[g] + b < w) colm[g] |= 0xffu << (8 * b);
#pragma unroll
        for (int q = 0; q < ND; q++) {
            unsigned m = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) if (x0[g] + (4 * q + b) / C < w) m |= 0xffu << (8 * b);
            pxm[g][q] = m;
            prev[g][q] = 0;
        }
#pragma unroll
        for (int i = 0; i < P; i++)
#pragma unroll
            for (int k = 0; k < EW; k++) { firE[g][i].d[k] = 0; iirE[g][i].d[k] = 0; }
    }
    auto on_bytes = [&](unsigned m, unsigned cm) {
        unsigned tt = (((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m) & 0x80808080u;
        tt = (tt - (tt >> 7)) | tt;
        return (tt ^ invm) & cm;
    };
    auto fir_entries = [&](const RowRegs<C, P> &r, unsigned cm, Ent (&e)[P]) {
        const unsigned on = on_bytes(r.m[0], cm);
        const unsigned on01 = on & 0x01010101u;
        if constexpr (C == 3) {
            const unsigned d0 = r.px[0] & __builtin_amdgcn_perm(0u, on, 0x01000000u);
            const unsigned d1 = r.px[1] & __builtin_amdgcn_perm(0u, on, 0x02020101u);
            const unsigned d2 = r.px[2] & __builtin_amdgcn_perm(0u, on, 0x03030302u);
            e[0].d[0] = __builtin_amdgcn_perm(d1, d0, 0x0c010c00u); e[0].d[1] = __builtin_amdgcn_perm(on01, d0, 0x0c040c02u);
            e[1].d[0] = __builtin_amdgcn_perm(d1, d0, 0x0c040c03u); e[1].d[1] = __builtin_amdgcn_perm(on01, d1, 0x0c050c01u);
            e[2].d[0] = __builtin_amdgcn_perm(d2, d1, 0x0c030c02u); e[2].d[1] = __builtin_amdgcn_perm(on01, d2, 0x0c060c00u);
            e[3].d[0] = __builtin_amdgcn_perm(d2, d2, 0x0c020c01u); e[3].d[1] = __builtin_amdgcn_perm(on01, d2, 0x0c070c03u);
        } else {
            const unsigned d0 = r.px[0] & on;
            e[0].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c040c00u);
            e[1].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c050c01u);
            e[2].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c060c02u);
            e[3].d[0] = __builtin_amdgcn_perm(on01, d0, 0x0c070c03u);
        }
    };
    auto iir_entries = [&](const unsigned (&o)[ND], Ent (&e)[P]) {
        if constexpr (C == 3) {
            e[0].d[0] = __builtin_amdgcn_perm(o[1], o[0], 0x0c010c00u); e[0].d[1] = __builtin_amdgcn_perm(0u, o[0], 0x0c0c0c02u);
            e[1].d[0] = __builtin_amdgcn_perm(o[1], o[0], 0x0c040c03u); e[1].d[1] = __builtin_amdgcn_perm(0u, o[1], 0x0c0c0c01u);
            e[2].d[0] = __builtin_amdgcn_perm(o[2], o[1], 0x0c030c02u); e[2].d[1] = __builtin_amdgcn_perm(0u, o[2], 0x0c0c0c00u);
            e[3].d[0] = __builtin_amdgcn_perm(o[2], o[2], 0x0c020c01u); e[3].d[1] = __builtin_amdgcn_perm(0u, o[2], 0x0c0c0c03u);
        } else {
            e[0].d[0] = o[0] & 0xffu; e[1].d[0] = (o[0] >> 8) & 0xffu; e[2].d[0] = (o[0] >> 16) & 0xffu; e[3].d[0] = o[0] >> 24;
        }
    };
    auto fir_apply = [&](int g, const RowRegs<C, P> &r, bool plus) {
        Ent e[P];
        fir_entries(r, colm[g], e);
#pragma unroll
        for (int i = 0; i < P; i++) { if (plus) eadd(firE[g][i], e[i]); else esub(firE[g][i], e[i]); }
    };
    // FIR rows [0, min(h, n-1)) enter before the loop; row y+n-1 enters at step y
    for (int yy = 0; yy < min(h, n - 1); yy++)
#pragma unroll
        for (int g = 0; g < G; g++) fir_apply(g, load_row_regs<C, P>(mask, mpitch, img, ipitch, yy, h, xl[g]), true);

    for (int y = 0; y < h; y++) {
        RowRegs<C, P> r_cur[G];
        // ---- this row's loads for both groups, then the vertical running sums ----
        {
            RowRegs<C, P> r_enter[G], r_leave[G];
            unsigned o_leave[G][ND];
#pragma unroll
            for (int g = 0; g < G; g++) {
                r_enter[g] = load_row_regs<C, P>(mask, mpitch, img, ipitch, y + n - 1, h, xl[g]);
                r_leave[g] = load_row_regs<C, P>(mask, mpitch, img, ipitch, y - n - 1, h, xl[g]);
                r_cur[g] = load_row_regs<C, P>(mask, mpitch, img, ipitch, y, h, xl[g]);
                gc_u32p p = (gc_u32p)(out + (size_t)min(max(y - n - 1, 0), h - 1) * opitch + (size_t)xl[g] * C);
#pragma unroll
                for (int q = 0; q < ND; q++) o_leave[g][q] = p[q];
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (y + n - 1 < h && n >= 1) fir_apply(g, r_enter[g], true);       // ye = min(h, y+n)
                if (y - n - 1 >= 0 && n >= 1) fir_apply(g, r_leave[g], false);     // ys = max(0, y-n)
                if (y >= 1 && n >= 1) {
                    Ent e[P];
                    iir_entries(prev[g], e);
#pragma unroll
                    for (int i = 0; i < P; i++) eadd(iirE[g][i], e[i]);
                }
                if (y - n - 1 >= 0 && n >= 1) {
                    unsigned ol[ND];
#pragma unroll
                    for (int q = 0; q < ND; q++) ol[q] = o_leave[g][q] & pxm[g][q];
                    Ent e[P];
                    iir_entries(ol, e);
#pragma unroll
                    for (int i = 0; i < P; i++) esub(iirE[g][i], e[i]);
                }
                if (x0[g] < wr)
#pragma unroll
                    for (int i = 0; i < P; i++) {
                        const int e = eidx(x0[g] + i);
                        if constexpr (EW == 2) {
                            *reinterpret_cast<uint2 *>(firA + e) = make_uint2(firE[g][i].d[0], firE[g][i].d[1]);
                            *reinterpret_cast<uint2 *>(iirA + e) = make_uint2(iirE[g][i].d[0], iirE[g][i].d[1]);
                        } else {
                            firA[e] = firE[g][i].d[0];
                            iirA[e] = iirE[g][i].d[0];
                        }
                    }
            }
        }
        const int ys = max(0, y - n);
        lds_barrier();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int xg = x0[g];
            const unsigned on_cur = on_bytes(r_cur[g].m[0], colm[g]);
            unsigned qd[ND];
#pragma unroll
            for (int q = 0; q < ND; q++) qd[q] = 0;
            if (__any(on_cur != colm[g])) {          // some pixel of this wave's 256 columns wants a quotient
                Ent aL, aR, aI;
#pragma unroll
                for (int k = 0; k < EW; k++) { aL.d[k] = 0; aR.d[k] = 0; aI.d[k] = 0; }
                for (int j = -n; j < 0; j++) { eadd(aL, lds_ld(firA, xg + j)); eadd(aI, lds_ld(iirA, xg + j)); }
                for (int j = 0; j < n; j++) {
                    if constexpr (NH == 2) eadd(aR, lds_ld(firA, xg + j)); else eadd(aL, lds_ld(firA, xg + j));
                }
#pragma unroll
                for (int i = 0; i < P; i++) {
                    const int x = xg + i;
                    const int xs = max(0, x - n);
                    int fsum[C], fcnt;
                    if constexpr (C == 3) {
                        fsum[0] = (int)(aL.d[0] & 0xffffu) + (int)(aI.d[0] & 0xffffu);
                        fsum[1] = (int)(aL.d[0] >> 16) + (int)(aI.d[0] >> 16);
                        fsum[2] = (int)(aL.d[1] & 0xffffu) + (int)(aI.d[1] & 0xffffu);
                        fcnt = (int)(aL.d[1] >> 16);
                        if constexpr (NH == 2) {
                            fsum[0] += (int)(aR.d[0] & 0xffffu); fsum[1] += (int)(aR.d[0] >> 16);
                            fsum[2] += (int)(aR.d[1] & 0xffffu); fcnt += (int)(aR.d[1] >> 16);
                        }
                    } else {
                        fsum[0] = (int)(aL.d[0] & 0xffffu) + (int)aI.d[0];
                        fcnt = (int)(aL.d[0] >> 16);
                        if constexpr (NH == 2) { fsum[0] += (int)(aR.d[0] & 0xffffu); fcnt += (int)(aR.d[0] >> 16); }
                    }
                    const int cnt = fcnt + (y - ys) * (x - xs);
                    const float rc = __builtin_amdgcn_rcpf((float)max(cnt, 1));   // 1 ulp: inside div_small's margin
                    const float hrc = 0.5f * rc;
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const unsigned q = (unsigned)__builtin_fmaf((float)fsum[c], rc, hrc);
                        const int jb = i * C + c;
                        qd[jb >> 2] |= q << (8 * (jb & 3));
                    }
                    if (i + 1 < P) {      // slide to pixel x+1
                        if constexpr (NH == 2) {
                            eadd(aL, firE[g][i]); esub(aL, lds_ld(firA, x - n));
                            eadd(aR, lds_ld(firA, x + n)); esub(aR, firE[g][i]);
                        } else {
                            eadd(aL, lds_ld(firA, x + n)); esub(aL, lds_ld(firA, x - n));
                        }
                        eadd(aI, iirE[g][i]); esub(aI, lds_ld(iirA, x - n));
                    }
                }
            }
            // masked pixels keep the image value (new_img = np.copy(img)), the others take the quotient
            unsigned res[ND];
            if constexpr (C == 3) {
                const unsigned e0 = __builtin_amdgcn_perm(0u, on_cur, 0x01000000u), e1 = __builtin_amdgcn_perm(0u, on_cur, 0x02020101u),
                               e2 = __builtin_amdgcn_perm(0u, on_cur, 0x03030302u);
                res[0] = ((r_cur[g].px[0] & e0) | (qd[0] & ~e0)) & pxm[g][0];
                res[1] = ((r_cur[g].px[1] & e1) | (qd[1] & ~e1)) & pxm[g][1];
                res[2] = ((r_cur[g].px[2] & e2) | (qd[2] & ~e2)) & pxm[g][2];
            } else {
                res[0] = ((r_cur[g].px[0] & on_cur) | (qd[0] & ~on_cur)) & pxm[g][0];
            }
            if (xg < w) {
                uint8_t *o = out + (size_t)y * opitch + (size_t)xg * C;
                if (xg + P <= w) {
#pragma unroll
                    for (int q = 0; q < ND; q++) ((g_u32p)o)[q] = res[q];
                } else {
                    const int nbytes = (w - xg) * C;
#pragma unroll
                    for (int j = 0; j < P * C; j++)
                        if (j < nbytes) ((g_u8p)(uintptr_t)o)[j] = (uint8_t)(res[j >> 2] >> (8 * (j & 3)));
                }
            }
#pragma unroll
            for (int q = 0; q < ND; q++) prev[g][q] = res[q];
        }
        lds_barrier();         // everyone is done reading the LDS rows
    }
}

template <int C, int NH>
__global__ __launch_bounds__(1024) void optimise_packed_wide_kernel(const OptJob *jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const OptJob J = jobs[blockIdx.x];
    if (J.n == 3) optimise_packed_wide_rows<C, 1, 3>(J, smem);
    else if (J.n == 10 && NH == 2) optimise_packed_wide_rows<C, 2, 10>(J, smem);
    else optimise_packed_wide_rows<C, NH, -1>(J, smem);
}

template <int C, int NH, int MAXT, bool DB>
__global__ __launch_bounds__(MAXT) void optimise_packed_kernel(const OptJob *jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const OptJob J = jobs[blockIdx.x];
    // wave-uniform dispatch on the job's n_size (the reference's two values get unrolled bodies) and on
    // the form of the mask (1-bpp rows from the denoiser when the job carries them)
    if (J.mbits) {
        if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, true>(J, smem);            // fg, mrc.py:413/415 (one accumulator: n <= 8)
        else if (J.n == 10 && NH == 2) optimise_packed_rows<C, 2, 10, DB, true>(J, smem);   // bg, mrc.py:447/449
        else optimise_packed_rows<C, NH, -1, DB, true>(J, smem);
    } else {
        if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, false>(J, smem);
        else if (J.n == 10 && NH == 2) optimise_packed_rows<C, 2, 10, DB, false>(J, smem);
        else optimise_packed_rows<C, NH, -1, DB, false>(J, smem);
    }
}

// grid = jobs * S workgroups: job = block / S, strip = block % S (left strips first)
template <int C, int NH, int MAXT>
__global__ __launch_bounds__(MAXT) void optimise_strip_kernel(const OptJob *jobs, StripInfo SI) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int job = blockIdx.x / SI.S, strip = blockIdx.x - job * SI.S;
    const OptJob J = jobs[job];
    // strips of up to 512 threads: the double-buffered LDS rows (one barrier per row) always fit; 1024 threads: single rows
    constexpr bool DB = MAXT <= 512;
    if (J.mbits) {
        if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, true, true>(J, smem, SI, job, strip);
        else if (J.n == 10 && NH == 2) optimise_packed_rows<C, 2, 10, DB, true, true>(J, smem, SI, job, strip);
        else optimise_packed_rows<C, NH, -1, DB, true, true>(J, smem, SI, job, strip);
    } else {
        if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, false, true>(J, smem, SI, job, strip);
        else if (J.n == 10 && NH == 2) optimise_packed_rows<C, 2, 10, DB, false, true>(J, smem, SI, job, strip);
        else optimise_packed_rows<C, NH, -1, DB, false, true>(J, smem, SI, job, strip);
    }
}

// Bands of every job of a launch (see OptBand).  One workgroup per job: row flags ("has an unselected pixel") from the
// 1-bpp mask rows, band starts (a flagged row with no flagged row among the n above it) and ends (none among the n below),
// one queue entry per band.  Queue layout: slots [0, njobs) hold at most one LONG band per job (more than half the page:
// the whole page of an fg layer) or an empty entry, so that the walkers start with the long chains; every other band is
// appended behind them (qctl[0] counts those).
__global__ __launch_bounds__(1024) void opt_bands_kernel(const OptJob *jobs, OptBand *q, unsigned *qctl, int njobs) {
    extern __shared__ unsigned bb[];                 // flag bits | start bits | end bits, nw words each
    const OptJob J = jobs[blockIdx.x];
    const int h = J.h, w = J.w, n = J.n, nw = (h + 31) >> 5;
    unsigned *fb = bb, *sb = bb + nw, *eb = bb + 2 * nw;
    const int t = threadIdx.x, NT = 1024;
    for (int i = t; i < 3 * nw; i += NT) bb[i] = 0;
    if (J.rowmap)                                     // bit y: row y is written by a band (the others stay image rows)
        for (int i = t; i < nw; i += NT) J.rowmap[i] = (J.rowflags && !J.invert) ? 0xffffffffu : 0u;
    if (t == 0) q[blockIdx.x] = OptBand{(int)blockIdx.x, 0, 0, 0, 0, {0, 0, 0}};
    __syncthreads();
    if (J.rowflags) {
        // the producer of the bit rows left one byte per row: an fg-like layer (unselected = clear bits) is one band anyway
        if (!J.invert) {
            if (t == 0) q[blockIdx.x] = OptBand{(int)blockIdx.x, 0, 0, h, h, {0, 0, 0}};
            return;
        }
        for (int y = t; y < h; y += NT)
            if (J.rowflags[y]) atomicOr(&fb[y >> 5], 1u << (y & 31));
    } else {
        // one wave per row, four rows in flight: lanes OR the row's words (rows of up to 4096 columns: two words per lane)
        const int wave = t >> 6, lane = t & 63, wpr = (w + 31) >> 5;
        const unsigned lastm = (w & 31) ? ((1u << (w & 31)) - 1u) : 0xffffffffu;
        const unsigned inv = J.invert ? 0u : 0xffffffffu;        // unselected = bit clear (or set, for an inverted mask)
        for (int y0 = 4 * wave; y0 < h; y0 += 4 * (NT / 64)) {
            unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int y = min(y0 + r, h - 1);
                for (int k = lane; k < wpr; k += 64) {
                    unsigned v = J.mbits[(size_t)y * J.mwpr + k] ^ inv;
                    if (k == wpr - 1) v &= lastm;
                    acc[r] |= v;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (y0 + r < h && __any(acc[r] != 0u) && lane == 0) atomicOr(&fb[(y0 + r) >> 5], 1u << ((y0 + r) & 31));
        }
    }
    __syncthreads();
    auto any_in = [&](const unsigned *b, int a, int e) {        // a flagged row in [a, e) (clipped)
        a = max(a, 0); e = min(e, h);
        for (int y = a; y < e;) {
            const int wi = y >> 5, lo = y & 31, cnt = min(32 - lo, e - y);
            const unsigned m = (cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u)) << lo;
            if (b[wi] & m) return true;
            y += cnt;
        }
        return false;
    };
    for (int y = t; y < h; y += NT) {
        if (!((fb[y >> 5] >> (y & 31)) & 1u)) continue;
        if (!any_in(fb, y - n, y)) atomicOr(&sb[y >> 5], 1u << (y & 31));
        if (!any_in(fb, y + 1, y + 1 + n)) atomicOr(&eb[y >> 5], 1u << (y & 31));
    }
    __syncthreads();
    auto next_bit = [&](const unsigned *b, int from) {          // first set bit at or after `from`, h if none
        for (int wi = from >> 5; wi < nw; wi++) {
            unsigned v = b[wi];
            if (wi == (from >> 5)) v &= 0xffffffffu << (from & 31);
            if (v) return min(h, wi * 32 + __builtin_ctz(v));
        }
        return h;
    };
    auto prev_bit = [&](const unsigned *b, int before) {        // last set bit before `before`, -1 if none
        for (int wi = (before - 1) >> 5; wi >= 0 && before > 0; wi--) {
            unsigned v = b[wi];
            if (wi == ((before - 1) >> 5) && ((before & 31) != 0)) v &= (1u << (before & 31)) - 1u;
            if (v) return wi * 32 + 31 - __builtin_clz(v);
        }
        return -1;
    };
    for (int y = t; y < h; y += NT) {
        if (!((sb[y >> 5] >> (y & 31)) & 1u)) continue;
        const int e = next_bit(eb, y);                           // the band's last flagged row (exists: ends and starts alternate)
        const int pe = prev_bit(eb, y);                          // the band before ends there
        const int ns = next_bit(sb, e + 1);
        OptBand b = {(int)blockIdx.x, pe + 1, y, e + 1, ns >= h ? h : e + 1, {0, 0, 0}};
        if (J.rowmap)
            for (int r = y; r <= e;) {
                const int lo = r & 31, cnt = min(32 - lo, e + 1 - r);
                atomicOr(&J.rowmap[r >> 5], (cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u)) << lo);
                r += cnt;
            }
        if ((b.y1 - b.y0) * 2 > h) q[blockIdx.x] = b;           // at most one such band per job
        else q[njobs + atomicAdd(&qctl[0], 1u)] = b;
    }
    if (t == 0 && next_bit(sb, 0) >= h)                          // no unselected pixel anywhere: the layer is a copy
        q[njobs + atomicAdd(&qctl[0], 1u)] = OptBand{(int)blockIdx.x, 0, 0, 0, h, {0, 0, 0}};
}

// a record at a wave-uniform address through the constant address space: scalar loads into scalar registers
template <class T>
__device__ __forceinline__ T load_uniform(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    typedef const unsigned __attribute__((address_space(4))) *c_u32p;
    const c_u32p src = (c_u32p)(uintptr_t)p;
    T v;
    unsigned *d = reinterpret_cast<unsigned *>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) d[i] = src[i];
    return v;
}

// Walkers: one workgroup per CU takes entries off the queue until it is empty.
// PAGES: the launch holds nothing but the pipeline's page-layers -- n = 3 jobs that are one whole-page band by
// construction (fg: row flags given, mask not inverted) and n = 10 jobs -- so only three instances of the row loop are
// compiled in instead of six.  (With six, the 1024-thread RGB kernel ran out of scalar registers and spilled 31 vector
// registers, some reloaded inside the row loops; with three, 7 launch-lifetime values are spilled at entry and reloaded
// once per queue entry, none in a row loop: tests/test_isa_checks.py.)
template <int C, int NH, int MAXT, bool DB, bool PAGES = false>
__global__ __launch_bounds__(MAXT) void optimise_band_kernel(const OptJob *jobs, const OptBand *q, unsigned *qctl, int njobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_idx;
    const int total = njobs + (int)__builtin_amdgcn_readfirstlane((int)qctl[0]);
    int zeroed_for_n = -1;                 // the LDS rows' pad entries are zero for this n (the layout depends on it)
    // Walker b starts with queue entry b -- consecutive workgroups sit on different XCDs, so the long chains at the front of
    // the queue (the fg pages) spread evenly over the eight L2s, as a launch of one workgroup per page would -- and takes
    // further entries off the shared counter.
    for (bool first = true;; first = false) {
        __syncthreads();                       // everyone has read s_idx of the previous round
        if (threadIdx.x == 0) s_idx = first ? (int)blockIdx.x : (int)(gridDim.x + atomicAdd(&qctl[1], 1u));
        __syncthreads();
        // wave-uniform by construction: make it so for the compiler too (scalar registers for the band, the job record and
        // every loop bound and row address derived from them; as lane values the row loop's control flow would be divergent)
        const int idx = __builtin_amdgcn_readfirstlane(s_idx);
        if (idx >= total) break;
        const OptBand B = load_uniform(q + idx);
        if (B.c1 <= B.c0) continue;            // an empty front slot
        const OptJob J = load_uniform(jobs + B.job);
        const bool zero = J.n != zeroed_for_n;
        zeroed_for_n = J.n;
        // A band that is (nearly) the whole page -- every fg layer -- takes the whole-page instance of the row loop: the
        // rows outside the band are copies either way, and without the band's bounds that loop is ~30 % leaner in
        // scalar instructions (measured: fg layers 5.2 -> 4.9 ms per 128 pages).
        const StripInfo NOSTRIP = StripInfo{1, 0, nullptr, 0, nullptr};
        // Only when the band is the job's ONLY band (it owns every row: c0 = 0, c1 = h): the whole-page instance
        // computes and stores all h rows, and another band of the same job walked at the same time by another workgroup
        // would have its rows written twice with no ordering (identical bytes, but not a pattern to rely on).
        const OptBand *bp = ((B.y1 - B.y0) * 10 >= J.h * 9 && B.c0 == 0 && B.c1 == J.h) ? nullptr : &B;
        if constexpr (PAGES) {
            static_assert(!PAGES || NH == 2, "page-layer launches: n = 3 and n = 10");
            if (bp) optimise_packed_rows<C, 2, 10, DB, true>(J, smem, NOSTRIP, 0, 0, bp, zero);
            else if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, true>(J, smem, NOSTRIP, 0, 0, nullptr, zero);
            else optimise_packed_rows<C, 2, 10, DB, true>(J, smem, NOSTRIP, 0, 0, nullptr, zero);
        } else if (bp) {
            if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, true>(J, smem, NOSTRIP, 0, 0, bp, zero);
            else if (J.n == 10 && NH == 2) optimise_packed_rows<C, 2, 10, DB, true>(J, smem, NOSTRIP, 0, 0, bp, zero);
            else optimise_packed_rows<C, NH, -1, DB, true>(J, smem, NOSTRIP, 0, 0, bp, zero);
        } else {
            if (J.n == 3) optimise_packed_rows<C, 1, 3, DB, true>(J, smem, NOSTRIP, 0, 0, nullptr, zero);
            else if (J.n == 10 && NH == 2) optimise_packed_rows<C, 2, 10, DB, true>(J, smem, NOSTRIP, 0, 0, nullptr, zero);
            else optimise_packed_rows<C, NH, -1, DB, true>(J, smem, NOSTRIP, 0, 0, nullptr, zero);
        }
    }
}

// Self-test of the quotient both optimise kernels use: (unsigned)fma((float)v, rcp(cnt), rcp(cnt)/2) against
// v / cnt for EVERY count the kernels can produce (1 .. 5120 = (2n)^2 + n^2 at n = 32) and every value
// 0 .. 255*cnt (a window of cnt bytes): ~3.3e9 pairs.
__global__ __launch_bounds__(256) void optimise_div_selftest_kernel(unsigned long long *bad) {
    const unsigned cnt = blockIdx.x + 1;                     // one workgroup per count
    const float rc = __builtin_amdgcn_rcpf((float)cnt), hrc = 0.5f * rc;
    const float qoff = __builtin_fmaf(rc, 0.5f, -0.5f);
    unsigned long long nbad = 0;
    for (unsigned v = threadIdx.x; v <= 255u * cnt; v += 256) {
        const unsigned q = (unsigned)__builtin_fmaf((float)v, rc, hrc);                     // the wide kernel's form
        if (q != v / cnt) nbad++;
        // the packed kernels' form: round-to-nearest-even byte conversion of (v + 0.5) / cnt - 0.5
        const unsigned qb = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)v, rc, qoff), 1, 0u) >> 8;
        if (qb != v / cnt) nbad++;
        if (div_small((int)v, rc) != v / cnt) nbad++;         // the unpacked kernel's form
    }
    if (nbad) atomicAdd(bad, nbad);
}

int optimise_div_selftest(mrchip_ctx *ctx, hipStream_t s, unsigned long long *d_bad) {
    HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
    hipLaunchKernelGGL(optimise_div_selftest_kernel, dim3(5120), dim3(256), 0, s, d_bad);
    HIP_TRY(hipGetLastError());
    return 0;
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
// Column strips (optimise_strip_kernel) when whole rows would leave most of the chip idle -- few page-layers -- or do
// not fit one workgroup (more than 4096 columns).  Returns 1 if it launched, 0 if the caller should go on, < 0 on error.
static int try_strips(mrchip_ctx *ctx, hipStream_t s, const OptJob *d_jobs, int njobs, int w, int h, int c, int n_max,
                      OptMail *mail, double alg) {
    const char *env = getenv("MRCHIP_OPT_STRIPS");          // 0: never, 128 / 256 / 512 / 1024: force that workgroup size
    const int mode = env ? atoi(env) : -1;
    if (!mail || mode == 0 || n_max > 11 || n_max < 1 || h >= 65536 || w < 64) return 0;
    const int cus = ctx->cus > 0 ? ctx->cus : 256;
    int T = 0;
    if (mode > 0) T = mode;
    else {
        // the narrowest strips that still give every workgroup a CU of its own; rows of more than 4096 columns do not
        // fit one workgroup of the packed kernel at all and always go in strips
        if (njobs * 2 <= cus || w > 4096)
            for (int cand : {128, 256, 512, 1024}) {
                const int sw = 4 * (cand - 2 * STRIP_HALO / 4);
                if ((long long)njobs * cdiv(w, sw) <= cus) { T = cand; break; }
            }
        if (T == 0 && w > 4096) T = 1024;
    }
    if (T == 0) return 0;
    const int sw = 4 * (T - 2 * STRIP_HALO / 4);
    const int S = cdiv(w, sw);
    if (S < 2) return 0;
    const size_t need = (size_t)njobs * (S - 1) * h * 4 * sizeof(u32x4) + 256;
    if (need > mail->bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        TRY(mail->buf.alloc(ctx, need + need / 4));
        mail->bytes = need + need / 4;
        mail->epoch = 0;
    }
    if (mail->epoch == 0 || mail->epoch >= 0xfffe) {       // fresh buffer, or the 16-bit epoch is about to repeat
        HIP_TRY(hipMemsetAsync(mail->buf.p, 0, mail->bytes, s));
        mail->epoch = 0;
    }
    mail->epoch++;
    StripInfo SI;
    SI.S = S; SI.sw = sw; SI.mail = mail->buf.as<u32x4>() + 16;           // (first 256 bytes: unused)
    SI.tagbase = mail->epoch << 16; SI.err = mail->err;                   // page-locked host word, see optmail_check
    const int nent = T * 4 + 2 * n_max;
    const size_t lds = (T <= 512 ? 2 : 1) * (size_t)(nent + nent / 4 + 1) * ((c == 3) ? 16 : 8);       // <= 512 threads: double-buffered rows
    const char *nm = c == 3 ? "optimise_rgb" : "optimise_gray";
#define OPT_STRIP(CC, NHH, MT)                                                                                   \
    do {                                                                                                        \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(optimise_strip_kernel<CC, NHH, MT>),         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        LAUNCH(ctx, s, nm, alg,                                                                                 \
               hipLaunchKernelGGL((optimise_strip_kernel<CC, NHH, MT>), dim3(njobs * S), dim3(T), lds, s, d_jobs, SI)); \
    } while (0)
    if (c == 3) {
        if (n_max <= 8) { if (T <= 512) OPT_STRIP(3, 1, 512); else OPT_STRIP(3, 1, 1024); }
        else { if (T <= 512) OPT_STRIP(3, 2, 512); else OPT_STRIP(3, 2, 1024); }
    } else {
        if (n_max <= 8) { if (T <= 512) OPT_STRIP(1, 1, 512); else OPT_STRIP(1, 1, 1024); }
        else { if (T <= 512) OPT_STRIP(1, 2, 512); else OPT_STRIP(1, 2, 1024); }
    }
#undef OPT_STRIP
    return 1;
}

OptMail::~OptMail() {
    if (err) (void)hipHostFree(err);
}

int optmail_check(OptMail *mail) {
    if (!mail || !mail->err) return 0;
    const unsigned e = *reinterpret_cast<volatile unsigned *>(mail->err);
    if (e) {
        *reinterpret_cast<volatile unsigned *>(mail->err) = 0;
        set_error("optimise: a strip gave up waiting for its left-hand neighbour's rows (hand-off timeout); the layers of this launch are invalid");
        return MRCHIP_E_HIP;
    }
    return 0;
}

int launch_optimise_jobs(mrchip_ctx *ctx, hipStream_t s, OptJob *h_jobs, OptJob *d_jobs, int njobs, int w, int h, int c,
                         int n_max, OptMail *mail) {
    if (c != 1 && c != 3) { set_error("optimise: channels must be 1 or 3"); return MRCHIP_E_ARG; }
    if (w <= 0 || h <= 0 || njobs <= 0) return 0;
    if (!mail) { set_error("optimise: no hand-off buffer"); return MRCHIP_E_ARG; }
    if (!mail->err) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&mail->err), 64, hipHostMallocMapped));
        *mail->err = 0;
    }
    int n_min = n_max;
    for (int i = 0; i < njobs; i++) n_min = std::min(n_min, h_jobs[i].n);
    if (n_max < 0 || n_max > 32) { set_error("optimise: n_size %d outside [0,32]", n_max); return MRCHIP_E_UNSUPPORTED; }
    // Whole rows of <= 4096 columns on one workgroup (four columns per thread), n <= 11: the band walkers
    // (optimise_band_kernel).  They read the mask at 1 bit per pixel: callers that only have bytes get a packed copy.
    // The whole-row geometry (opt_geometry: it refuses rows that do not fit one workgroup's LDS) is only asked for once
    // the column strips have declined the launch: rows of more than 4096 columns always go in strips, whatever their width.
    static const bool no_bands = getenv("MRCHIP_OPT_BANDS") && atoi(getenv("MRCHIP_OPT_BANDS")) == 0;
    const bool bands = !no_bands && cdiv(w, 4) <= 1024 && n_max <= 11 && n_min >= 1 && h < 65536;
    if (bands) {
        const int wpr = cdiv(w, 32);
        const size_t per = (size_t)wpr * h * sizeof(unsigned);
        size_t missing = 0;
        for (int i = 0; i < njobs; i++) if (!h_jobs[i].mbits) missing++;
        if (missing) {
            if (mail->bits_bytes < missing * per) {
                HIP_TRY(hipStreamSynchronize(s));
                TRY(mail->bits.alloc(ctx, missing * per));
                mail->bits_bytes = missing * per;
            }
            size_t k = 0;
            for (int i = 0; i < njobs; i++) {
                OptJob &j = h_jobs[i];
                if (j.mbits) continue;
                // jobs that share a byte mask (fg and bg of one page) share the packed copy
                unsigned *dst = nullptr;
                for (int q = 0; q < i && !dst; q++)
                    if (h_jobs[q].mask == j.mask && h_jobs[q].mpitch == j.mpitch) dst = const_cast<unsigned *>(h_jobs[q].mbits);
                if (!dst) {
                    dst = reinterpret_cast<unsigned *>(mail->bits.as<unsigned char>() + (k++) * per);
                    TRY(launch_pack_bits(ctx, s, j.mask, j.mpitch, w, h, dst, wpr));
                }
                j.mbits = dst; j.mwpr = wpr;
            }
        }
    }
    // skip_copy is honoured by the band walkers only: the records go up without it first (the strips may take the launch)
    std::vector<int> want_skip(njobs);
    bool any_skip = false;
    for (int i = 0; i < njobs; i++) {
        want_skip[i] = bands && h_jobs[i].skip_copy;
        any_skip = any_skip || want_skip[i];
        h_jobs[i].skip_copy = 0; h_jobs[i].rowmap = nullptr;
    }
    TRY(upload_1d(s, d_jobs, h_jobs, (size_t)njobs * sizeof(OptJob)));
    {
        const int st = try_strips(ctx, s, d_jobs, njobs, w, h, c, n_max, mail, (1.0 + 2.0 * c) * w * h * njobs);
        if (st != 0) return st < 0 ? st : 0;
    }
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
    // packed kernel: entries only for real columns; double-buffer the rows when twice that fits
    const int wr = std::min(g.T * 4, (w + 3) & ~3);
    const int pnent = wr + 2 * n_max;
    const size_t plds1 = (size_t)(pnent + pnent / 4 + 1) * ((c == 3) ? 16 : 8);
    // double-buffered LDS rows (row y publishes into buffer y & 1: one barrier per row instead of two) whenever twice the
    // rows fit: 256 page-layers of 4000 columns 6.06 -> 5.58 ms (round 1 measured no gain: the kernel then had more
    // VALU work per row to hide the second barrier behind).
    const bool db = 2 * plds1 <= 160 * 1024;
    const size_t plds = db ? 2 * plds1 : plds1;
#define OPT_PACKED2(CC, NHH, MT, DBB, NAME)                                                             \
    do {                                                                                                \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(optimise_packed_kernel<CC, NHH, MT, DBB>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));            \
        LAUNCH(ctx, s, NAME, alg,                                                                       \
               hipLaunchKernelGGL((optimise_packed_kernel<CC, NHH, MT, DBB>), dim3(njobs), dim3(g.T), plds, s, d_jobs)); \
    } while (0)
#define OPT_PACKED(CC, NHH, MT, NAME)                                                                    \
    do { if (db) OPT_PACKED2(CC, NHH, MT, true, NAME); else OPT_PACKED2(CC, NHH, MT, false, NAME); } while (0)
    // wide rows: two groups of 4 columns per thread, LDS rows of one entry per column must still fit
    const int wwr = std::min(g.T * 8, (w + 3) & ~3), wnent = wwr + 2 * n_max;
    const size_t wlds = (size_t)(wnent + wnent / 4 + 1) * ((c == 3) ? 16 : 8);
    const bool wide_ok = wlds <= 160 * 1024;
    if (bands) {
        // queue: njobs front slots + one entry per band (a band and its gap take more than n_min rows) + control words
        const size_t cap = (size_t)njobs * (2 + h / (n_min + 1));
        const size_t nw = (size_t)cdiv(h, 32);
        const size_t need = 256 + cap * sizeof(OptBand) + (size_t)njobs * nw * sizeof(unsigned);
        if (mail->bandq_bytes < need) {
            HIP_TRY(hipStreamSynchronize(s));
            TRY(mail->bandq.alloc(ctx, need));
            mail->bandq_bytes = need;
        }
        unsigned *qctl = mail->bandq.as<unsigned>();
        OptBand *q = reinterpret_cast<OptBand *>(mail->bandq.as<unsigned char>() + 256);
        if (any_skip) {
            // the row maps sit behind the queue (job k's at k * nw words): a consumer of consecutive jobs strides by nw
            unsigned *maps = reinterpret_cast<unsigned *>(mail->bandq.as<unsigned char>() + 256 + cap * sizeof(OptBand));
            for (int i = 0; i < njobs; i++)
                if (want_skip[i]) { h_jobs[i].skip_copy = 1; h_jobs[i].rowmap = maps + (size_t)i * nw; }
            TRY(upload_1d(s, d_jobs, h_jobs, (size_t)njobs * sizeof(OptJob)));
        }
        HIP_TRY(hipMemsetAsync(qctl, 0, 8, s));
        LAUNCH(ctx, s, "optimise_bands", 0.0,
               hipLaunchKernelGGL(opt_bands_kernel, dim3(njobs), dim3(1024), 3 * cdiv(h, 32) * sizeof(unsigned), s, d_jobs, q, qctl, njobs));
        const int cus = ctx->cus > 0 ? ctx->cus : 256;
        // one walker per CU when the rows need more than half a CU's threads or LDS, else two
        const int per_cu = (g.T <= 512 && 2 * plds <= 160 * 1024) ? 2 : 1;
        const int walkers = std::min<long long>((long long)cap, (long long)cus * per_cu);
        // page-layer launches (every n = 3 job a whole-page band by construction, everything else n = 10): the kernel
        // instance with three row loops instead of six
        static const bool no_pages = getenv("MRCHIP_OPT_PAGES") && atoi(getenv("MRCHIP_OPT_PAGES")) == 0;
        bool pages_only = n_max == 10 && !no_pages;
        for (int i = 0; i < njobs && pages_only; i++)
            pages_only = (h_jobs[i].n == 3 && h_jobs[i].rowflags && !h_jobs[i].invert) || h_jobs[i].n == 10;
#define OPT_BAND3(CC, NHH, MT, DBB, PG, NAME)                                                            \
    do {                                                                                                \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(optimise_band_kernel<CC, NHH, MT, DBB, PG>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));            \
        LAUNCH(ctx, s, NAME, alg,                                                                       \
               hipLaunchKernelGGL((optimise_band_kernel<CC, NHH, MT, DBB, PG>), dim3(walkers), dim3(g.T), plds, s, d_jobs, q, qctl, njobs)); \
    } while (0)
#define OPT_BAND2(CC, NHH, MT, DBB, NAME)                                                                \
    do {                                                                                                \
        if constexpr (NHH == 2) { if (pages_only) OPT_BAND3(CC, NHH, MT, DBB, true, NAME); else OPT_BAND3(CC, NHH, MT, DBB, false, NAME); } \
        else OPT_BAND3(CC, NHH, MT, DBB, false, NAME);                                                   \
    } while (0)
#define OPT_BAND(CC, NHH, MT, NAME)                                                                      \
    do { if (db) OPT_BAND2(CC, NHH, MT, true, NAME); else OPT_BAND2(CC, NHH, MT, false, NAME); } while (0)
        if (c == 3) {
            if (n_max <= 8) { if (g.T <= 512) OPT_BAND(3, 1, 512, "optimise_rgb"); else OPT_BAND(3, 1, 1024, "optimise_rgb"); }
            else { if (g.T <= 512) OPT_BAND(3, 2, 512, "optimise_rgb"); else OPT_BAND(3, 2, 1024, "optimise_rgb"); }
        } else {
            if (n_max <= 8) { if (g.T <= 512) OPT_BAND(1, 1, 512, "optimise_gray"); else OPT_BAND(1, 1, 1024, "optimise_gray"); }
            else { if (g.T <= 512) OPT_BAND(1, 2, 512, "optimise_gray"); else OPT_BAND(1, 2, 1024, "optimise_gray"); }
        }
#undef OPT_BAND
#undef OPT_BAND2
#undef OPT_BAND3
    } else if (g.P == 4 && n_max <= 11) {
        // 16-bit lane capacity: one FIR accumulator up to n=8, two halves up to n=11
        if (c == 3) {
            if (n_max <= 8) { if (g.T <= 512) OPT_PACKED(3, 1, 512, "optimise_rgb"); else OPT_PACKED(3, 1, 1024, "optimise_rgb"); }
            else { if (g.T <= 512) OPT_PACKED(3, 2, 512, "optimise_rgb"); else OPT_PACKED(3, 2, 1024, "optimise_rgb"); }
        } else {
            if (n_max <= 8) { if (g.T <= 512) OPT_PACKED(1, 1, 512, "optimise_gray"); else OPT_PACKED(1, 1, 1024, "optimise_gray"); }
            else { if (g.T <= 512) OPT_PACKED(1, 2, 512, "optimise_gray"); else OPT_PACKED(1, 2, 1024, "optimise_gray"); }
        }
    } else if (g.P == 8 && n_max <= 11 && wide_ok) {
        // 4097..8160 columns: two column groups per thread in the packed scheme
#define OPT_WIDE(CC, NHH, NAME)                                                                          \
    do {                                                                                                \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(optimise_packed_wide_kernel<CC, NHH>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));            \
        LAUNCH(ctx, s, NAME, alg,                                                                       \
               hipLaunchKernelGGL((optimise_packed_wide_kernel<CC, NHH>), dim3(njobs), dim3(g.T), wlds, s, d_jobs)); \
    } while (0)
        if (c == 3) { if (n_max <= 8) OPT_WIDE(3, 1, "optimise_rgb"); else OPT_WIDE(3, 2, "optimise_rgb"); }
        else { if (n_max <= 8) OPT_WIDE(1, 1, "optimise_gray"); else OPT_WIDE(1, 2, "optimise_gray"); }
#undef OPT_WIDE
    } else if (c == 3) OPT_PICK(3, "optimise_rgb");
    else OPT_PICK(1, "optimise_gray");
#undef OPT_PACKED
#undef OPT_PACKED2
#undef OPT_PICK
#undef OPT_LAUNCH
    return 0;
}

}  // namespace mrchip
