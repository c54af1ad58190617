// internetarchivepdf/grayconvert.py:38-66 `special_gray_convert` (recode.py:362, the --grayscale-pdf "special" option;
// SURVEY.md 8f rank 4) for gfx950: two passes over the interleaved RGB page, both HBM-bound byte work.
//
//   rgb_stats     grayconvert.py:41-44: min / max / sum / sum of squares of every channel, exact integers (the host
//                 turns them into the reference's mean and std; 3*w*h bytes read).  One lane takes 4 pixels = three
//                 aligned dwords; `v_perm_b32` gathers each channel's four bytes into two 16-bit pairs, so that the
//                 sums are `v_dot2_u32_u16` (pair . (1,1) and pair . pair) and the extremes `v_pk_min_u16` /
//                 `v_pk_max_u16`: 10 VALU instructions per channel and group, no byte extraction.
//   rgb_level_hsl grayconvert.py:56-66: per channel `level_arr` (a function of the byte: a 256-entry table the HOST
//                 builds with the reference's own numpy expressions), then skimage's rgb2hsv and l = V (1 - S / 2),
//                 uint8(l * 255) -- a function of the pixel's (max, min) only: V = max / 255 and S = (max - min) / max in
//                 float64, so a 256 x 256 byte table, again built on the host in the reference's operation order
//                 and checked against scikit-image over all 65536 pairs (tests/golden/grayconvert.npz).  Both tables
//                 live in LDS (66 304 bytes); workgroups are persistent over row chunks so that the staging is paid
//                 once per workgroup.  4*w*h algorithmic bytes (3 read + 1 written per pixel).
// Bit-exact by construction: no floating point on the device.
#include "mrchip_internal.h"

namespace mrchip {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned wave_min(unsigned v) { for (int o = 32; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ unsigned wave_max(unsigned v) { for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// rows [y0, y1) of one page per workgroup-row (blockIdx.y), groups of 4 pixels strided over the workgroup's lanes
__global__ __launch_bounds__(256) void rgb_stats_kernel(const uint8_t *rgb, int pitch, int w, int h, int rows_per_block,
                                                        RgbStats *out) {
    const int y0 = blockIdx.y * rows_per_block, y1 = min(h, y0 + rows_per_block);
    const int ngroups = w >> 2;                        // whole groups of 4 pixels; the last w & 3 pixels go bytewise
    // 16-bit lanes: running extremes as pairs, sums per row in 32 bits (a row adds at most 2 * 65025 per dot2), 64-bit totals
    u16x2 mn[3][2], mx[3][2];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 2; k++) { mn[c][k] = u16x2{255, 255}; mx[c][k] = u16x2{0, 0}; }
    unsigned long long sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    const u16x2 ones = u16x2{1, 1};
    for (int y = y0; y < y1; y++) {
        const uint8_t *row = rgb + (size_t)y * pitch;
        unsigned rs[3] = {0, 0, 0}, rq[3] = {0, 0, 0};
        for (int g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
            const unsigned *p = reinterpret_cast<const unsigned *>(row + (size_t)g * 12);     // 12 g: dword aligned (pitch % 64 == 0)
            const unsigned a = p[0], b = p[1], c = p[2];
            // bytes: a = R0 G0 B0 R1, b = G1 B1 R2 G2, c = B2 R3 G3 B3.  v_perm_b32(hi, lo, sel): selector byte k picks byte
            // sel_k of {lo: 0..3, hi: 4..7}, 0x0c = the constant 0 -> two zero-extended 16-bit lanes per instruction
            const unsigned pr[3][2] = {
                {__builtin_amdgcn_perm(0u, a, 0x0c030c00u), __builtin_amdgcn_perm(c, b, 0x0c050c02u)},     // R0 R1 | R2 R3
                {__builtin_amdgcn_perm(b, a, 0x0c040c01u), __builtin_amdgcn_perm(c, b, 0x0c060c03u)},     // G0 G1 | G2 G3
                {__builtin_amdgcn_perm(b, a, 0x0c050c02u), __builtin_amdgcn_perm(0u, c, 0x0c030c00u)}};   // B0 B1 | B2 B3
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const u16x2 v = __builtin_bit_cast(u16x2, pr[ch][k]);
                    mn[ch][k] = __builtin_elementwise_min(mn[ch][k], v);
                    mx[ch][k] = __builtin_elementwise_max(mx[ch][k], v);
                    rs[ch] = __builtin_amdgcn_udot2(v, ones, rs[ch], false);
                    rq[ch] = __builtin_amdgcn_udot2(v, v, rq[ch], false);
                }
        }
        if (blockIdx.x == 0 && threadIdx.x < (unsigned)(w & 3)) {           // the row's last 1..3 pixels
            const uint8_t *px = row + (size_t)(ngroups * 4 + threadIdx.x) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const unsigned short v = px[ch];
                mn[ch][0] = __builtin_elementwise_min(mn[ch][0], u16x2{v, v});
                mx[ch][0] = __builtin_elementwise_max(mx[ch][0], u16x2{v, v});
                rs[ch] += v; rq[ch] += (unsigned)v * v;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { sum[ch] += rs[ch]; sq[ch] += rq[ch]; }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const u16x2 m2 = __builtin_elementwise_min(mn[ch][0], mn[ch][1]), M2 = __builtin_elementwise_max(mx[ch][0], mx[ch][1]);
        const unsigned m = wave_min(min((unsigned)m2.x, (unsigned)m2.y)), M = wave_max(max((unsigned)M2.x, (unsigned)M2.y));
        const unsigned long long s = wave_sum64(sum[ch]), q = wave_sum64(sq[ch]);
        if (lane == 0) {
            atomicMin(&out->mn[ch], m);
            atomicMax(&out->mx[ch], M);
            atomicAdd(&out->sum[ch], s);
            atomicAdd(&out->sumsq[ch], q);
        }
    }
}

// tables: lut[3][256] (level_arr per channel) then hsl[256][256] (index: max * 256 + min)
constexpr int GRAY_TABLE_BYTES = 3 * 256 + 256 * 256;

__global__ __launch_bounds__(256) void rgb_level_hsl_kernel(const uint8_t *rgb, int rgb_pitch, uint8_t *gray, int gray_pitch, int w,
                                                            int h, const uint8_t *tables) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *src = reinterpret_cast<const u32x4 *>(tables);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
        for (int i = threadIdx.x; i < GRAY_TABLE_BYTES / 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const unsigned char *lr = lds, *lg = lds + 256, *lb = lds + 512, *hsl = lds + 768;
    auto one = [&](unsigned r, unsigned g, unsigned b) -> unsigned {
        const unsigned R = lr[r], G = lg[g], B = lb[b];
        const unsigned mx = max(R, max(G, B)), mn = min(R, min(G, B));      // v_max3_u32 / v_min3_u32
        return hsl[mx * 256u + mn];
    };
    const int groups = (w + 3) >> 2;
    // persistent workgroups: (row, block of 256 groups) pairs in a grid-stride loop
    const int xblocks = (groups + 255) >> 8;
    const long long total = (long long)xblocks * h;
    for (long long t = blockIdx.x; t < total; t += gridDim.x) {
        const int y = (int)(t / xblocks), x4 = ((int)(t % xblocks) * 256 + (int)threadIdx.x) * 4;
        if (x4 >= w) continue;
        const uint8_t *row = rgb + (size_t)y * rgb_pitch + (size_t)x4 * 3;
        uint8_t *out = gray + (size_t)y * gray_pitch + x4;
        if (x4 + 4 <= w) {
            const unsigned *p = reinterpret_cast<const unsigned *>(row);
            const unsigned a = p[0], b = p[1], c = p[2];
            const unsigned l0 = one(a & 0xff, (a >> 8) & 0xff, (a >> 16) & 0xff);
            const unsigned l1 = one(a >> 24, b & 0xff, (b >> 8) & 0xff);
            const unsigned l2 = one((b >> 16) & 0xff, b >> 24, c & 0xff);
            const unsigned l3 = one((c >> 8) & 0xff, (c >> 16) & 0xff, c >> 24);
            *reinterpret_cast<unsigned *>(out) = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
        } else {
            for (int i = 0; x4 + i < w; i++) out[i] = (uint8_t)one(row[3 * i], row[3 * i + 1], row[3 * i + 2]);
        }
    }
}

int launch_rgb_stats(mrchip_ctx *ctx, hipStream_t s, const uint8_t *rgb, int pitch, int w, int h, void *d_stats) {
    RgbStats init;
    memset(&init, 0, sizeof(init));
    for (int c = 0; c < 3; c++) init.mn[c] = 255u;
    // (the 80-byte initial value travels inside the memcpy node's own staging: `init` may leave scope at once)
    TRY(upload_1d(s, d_stats, &init, sizeof(init)));
    HIP_TRY(hipStreamSynchronize(s));
    const int rows_per_block = 16;
    const int gx = std::max(1, std::min(cdiv(w >> 2, 256), 8));
    dim3 grid(gx, cdiv(h, rows_per_block));
    LAUNCH(ctx, s, "rgb_stats", 3.0 * w * h,
           hipLaunchKernelGGL(rgb_stats_kernel, grid, dim3(256), 0, s, rgb, pitch, w, h, rows_per_block, (RgbStats *)d_stats));
    return 0;
}

int launch_rgb_level_hsl(mrchip_ctx *ctx, hipStream_t s, const uint8_t *rgb, int rgb_pitch, uint8_t *gray, int gray_pitch, int w,
                         int h, const uint8_t *d_tables) {
    HIP_TRY(hipFuncSetAttribute((const void *)rgb_level_hsl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GRAY_TABLE_BYTES));
    const long long total = (long long)cdiv(cdiv(w, 4), 256) * h;
    const int cus = ctx->cus > 0 ? ctx->cus : 256;
    const int grid = (int)std::max<long long>(1, std::min<long long>(total, (long long)cus * 2));     // two 66 KB workgroups per CU
    LAUNCH(ctx, s, "rgb_level_hsl", 4.0 * w * h,
           hipLaunchKernelGGL(rgb_level_hsl_kernel, dim3(grid), dim3(256), GRAY_TABLE_BYTES, s, rgb, rgb_pitch, gray, gray_pitch, w, h,
                              d_tables));
    return 0;
}

}  // namespace mrchip
