// Measurement tool, not part of the library: time of one host -> device -> host round trip of a PAGEABLE h x w byte image, as a true 2D copy into
// a pitched device image (hipMemcpy2DAsync, what upload_2d / download_2d of ctx.hip do) and as one contiguous copy (hipMemcpyAsync).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/pageable_copy_rates.hip -o tools/ubench/bin/pageable_copy_rates && tools/ubench/bin/pageable_copy_rates
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned char *dev = nullptr;
    CK(hipMalloc(&dev, 256u << 20));
    unsigned char *pin = nullptr;
    CK(hipHostMalloc((void **)&pin, 64u << 20, hipHostMallocDefault));
    const int shapes[][2] = {{60, 60}, {360, 744}, {500, 900}, {1600, 1200}, {3000, 4000}, {3000, 12000}};
    for (auto &sh : shapes) {
        const int h = sh[0], w = sh[1], pitch = (w + 255) & ~255;
        std::vector<unsigned char> a((size_t)h * w, 7), b((size_t)h * w);
        const int reps = h * w > 4000000 ? 5 : 50;
        auto run = [&](int mode) -> double {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) {
                if (mode == 0) {            // pitched device image, pageable host: 2D both ways
                    (void)hipMemcpy2DAsync(dev, pitch, a.data(), w, w, h, hipMemcpyHostToDevice, s);
                    (void)hipMemcpy2DAsync(b.data(), w, dev, pitch, w, h, hipMemcpyDeviceToHost, s);
                } else if (mode == 1) {     // contiguous both sides, pageable host
                    (void)hipMemcpyAsync(dev, a.data(), (size_t)h * w, hipMemcpyHostToDevice, s);
                    (void)hipMemcpyAsync(b.data(), dev, (size_t)h * w, hipMemcpyDeviceToHost, s);
                } else if (mode == 2) {     // pitched device image, PINNED host: 2D both ways
                    (void)hipMemcpy2DAsync(dev, pitch, pin, w, w, h, hipMemcpyHostToDevice, s);
                    (void)hipMemcpy2DAsync(pin, w, dev, pitch, w, h, hipMemcpyDeviceToHost, s);
                } else {                    // host memcpy into pinned + contiguous copies
                    memcpy(pin, a.data(), (size_t)h * w);
                    (void)hipMemcpyAsync(dev, pin, (size_t)h * w, hipMemcpyHostToDevice, s);
                    (void)hipMemcpyAsync(pin, dev, (size_t)h * w, hipMemcpyDeviceToHost, s);
                    (void)hipStreamSynchronize(s);
                    memcpy(b.data(), pin, (size_t)h * w);
                }
                (void)hipStreamSynchronize(s);
            }
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps * 1e3;
        };
        run(0); run(1); run(2); run(3);
        printf("%5d x %5d: 2D pageable %8.3f ms   1D pageable %8.3f ms   2D pinned %8.3f ms   memcpy + 1D pinned %8.3f ms\n", h, w, run(0), run(1), run(2), run(3));
    }
    return 0;
}
