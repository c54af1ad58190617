// PIL Image.convert('L') for RGB input (mrc.py:361): ITU-R 601-2 luma in 16.16
// fixed point, L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16  (SURVEY.md a10).
// One lane converts 4 pixels: three aligned dword loads (12 B) -> one dword store.
// Algorithmic bytes: (3+1)*w*h.
#include "mrchip_internal.h"

namespace mrchip {

__device__ __forceinline__ unsigned luma1(unsigned r, unsigned g, unsigned b) {
    return (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16;
}

__global__ __launch_bounds__(256) void luma601_kernel(const uint8_t *rgb, int rgb_pitch, size_t rgb_stride,
                                                      uint8_t *gray, int gray_pitch, size_t gray_stride, int w, int h) {
    const int y = blockIdx.y;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= w) return;
    rgb += (size_t)blockIdx.z * rgb_stride;
    gray += (size_t)blockIdx.z * gray_stride;
    const uint8_t *row = rgb + (size_t)y * rgb_pitch + (size_t)x4 * 3;
    uint8_t *out = gray + (size_t)y * gray_pitch + x4;
    if (x4 + 4 <= w) {
        const unsigned *p = reinterpret_cast<const unsigned *>(row);   // 12*x aligned: pitch%64==0
        unsigned a = p[0], b = p[1], c = p[2];
        unsigned l0 = luma1(a & 0xff, (a >> 8) & 0xff, (a >> 16) & 0xff);
        unsigned l1 = luma1(a >> 24, b & 0xff, (b >> 8) & 0xff);
        unsigned l2 = luma1((b >> 16) & 0xff, b >> 24, c & 0xff);
        unsigned l3 = luma1((c >> 8) & 0xff, (c >> 16) & 0xff, c >> 24);
        *reinterpret_cast<unsigned *>(out) = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
    } else {
        for (int i = 0; x4 + i < w; i++) out[i] = (uint8_t)luma1(row[3 * i], row[3 * i + 1], row[3 * i + 2]);
    }
}

int launch_luma601(mrchip_ctx *ctx, hipStream_t s, Plane rgb, Plane gray, int w, int h, int npages) {
    dim3 grid(cdiv(cdiv(w, 4), 256), h, npages);
    LAUNCH(ctx, s, "luma601", 4.0 * w * h * npages,
           hipLaunchKernelGGL(luma601_kernel, grid, dim3(256), 0, s, rgb.p, rgb.pitch, rgb.stride, gray.p, gray.pitch,
                              gray.stride, w, h));
    return 0;
}

}  // namespace mrchip
