// Measurement / diagnosis tool, not part of the library: does hipMemcpy2DAsync between PAGEABLE host arrays and pitched device images -- the
// way the single-call entry points of api_kernels.hip stage a numpy array (upload_2d / download_2d, ctx.hip) -- return every byte, every
// time?  Random shapes like the fuzz sweep's (8..500 rows, 8..900 columns, device pitch rounded up), fresh malloc'ed host arrays, a
// non-blocking stream, upload -> (optional device-side touch) -> download -> memcmp.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/pageable_copy2d.hip -o tools/ubench/bin/pageable_copy2d && tools/ubench/bin/pageable_copy2d [seconds] [seed]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void invert_kernel(unsigned char *p, int pitch, int w, int h) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < w && y < h) p[(size_t)y * pitch + x] ^= 0xff;
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 30.0;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
    std::mt19937 rng(seed);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned char *dev = nullptr;
    CK(hipMalloc(&dev, 64u << 20));
    const auto t0 = std::chrono::steady_clock::now();
    long long n = 0, bad = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        const int h = 8 + (int)(rng() % 492), w = 8 + (int)(rng() % 892);
        const int pitch = (w + 255) & ~255;
        const size_t off = (size_t)(rng() % 4096) * 256;                      // the device block moves around like a cached allocation
        std::vector<unsigned char> a((size_t)h * w), b((size_t)h * w, 0x5a);
        for (auto &v : a) v = (unsigned char)rng();
        const bool touch = rng() & 1;
        CK(hipMemcpy2DAsync(dev + off, pitch, a.data(), w, w, h, hipMemcpyHostToDevice, s));
        if (touch) hipLaunchKernelGGL(invert_kernel, dim3((w + 255) / 256, h), dim3(256), 0, s, dev + off, pitch, w, h);
        CK(hipMemcpy2DAsync(b.data(), w, dev + off, pitch, w, h, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (touch) for (auto &v : b) v ^= 0xff;
        if (memcmp(a.data(), b.data(), a.size()) != 0) {
            size_t first = 0, cnt = 0;
            for (size_t i = 0; i < a.size(); i++) if (a[i] != b[i]) { if (!cnt) first = i; cnt++; }
            printf("MISMATCH case %lld: %d x %d touch %d: %zu bytes differ, first at row %zu column %zu\n", n, h, w, (int)touch, cnt, first / w, first % w);
            bad++;
        }
        n++;
    }
    printf("pageable 2D copies: %lld round trips in %.0f s, seed %u, %lld mismatches\n", n, secs, seed, bad);
    return bad ? 1 : 0;
}
