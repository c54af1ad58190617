// probe: host<->device copy rates one direction at a time and both at once (page-locked memory, two streams):
// how much of the link's two directions a pipeline that uploads and downloads concurrently can get.
//   hipcc --offload-arch=gfx950 -O3 -o pcie_duplex pcie_duplex.hip && ./pcie_duplex [MiB per copy] [copies]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 16;
    const size_t n = mib << 20;
    void *hin, *hout, *din, *dout;
    CK(hipHostMalloc(&hin, n, hipHostMallocDefault));
    CK(hipHostMalloc(&hout, n, hipHostMallocDefault));
    memset(hin, 1, n); memset(hout, 2, n);
    CK(hipMalloc(&din, n)); CK(hipMalloc(&dout, n));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; mode++) {
        for (int warm = 0; warm < 2; warm++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int r = 0; r < reps; r++) {
                if (mode != 1) CK(hipMemcpyAsync(din, hin, n, hipMemcpyHostToDevice, s1));
                if (mode != 0) CK(hipMemcpyAsync(hout, dout, n, hipMemcpyDeviceToHost, s2));
            }
            CK(hipDeviceSynchronize());
            const double dt = now() - t0;
            if (warm) {
                const double gb = (double)n * reps / 1e9;
                if (mode == 0) printf("H2D alone      : %6.1f GB/s\n", gb / dt);
                if (mode == 1) printf("D2H alone      : %6.1f GB/s\n", gb / dt);
                if (mode == 2) printf("both at once   : %6.1f GB/s each way, %6.1f GB/s together\n", gb / dt, 2 * gb / dt);
            }
        }
    }
    // smaller copies (a page of 36 MB) issued back to back on each stream
    const size_t pg = 36u << 20;
    CK(hipDeviceSynchronize());
    const double t0 = now();
    const int k = (int)(n / pg);
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < k; i++) {
            CK(hipMemcpyAsync((char *)din + i * pg, (char *)hin + i * pg, pg, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync((char *)hout + i * pg, (char *)dout + i * pg, pg, hipMemcpyDeviceToHost, s2));
        }
    CK(hipDeviceSynchronize());
    const double dt = now() - t0;
    printf("36 MiB copies  : %6.1f GB/s each way\n", (double)pg * k * 4 / 1e9 / dt);
    // pitched device images (rows of 12000 bytes at a pitch of 12096), tight host arrays: what the page uploads / layer
    // downloads are
    {
        const size_t wb = 12000, dp = 12096, rows = 3000;
        const int pages = (int)(n / (dp * rows));
        for (int mode = 0; mode < 3; mode++) {
            CK(hipDeviceSynchronize());
            const double t1 = now();
            for (int r = 0; r < 4; r++)
                for (int i = 0; i < pages; i++) {
                    if (mode != 1) CK(hipMemcpy2DAsync((char *)din + i * dp * rows, dp, (char *)hin + i * wb * rows, wb, wb, rows, hipMemcpyHostToDevice, s1));
                    if (mode != 0) CK(hipMemcpy2DAsync((char *)hout + i * wb * rows, wb, (char *)dout + i * dp * rows, dp, wb, rows, hipMemcpyDeviceToHost, s2));
                }
            CK(hipDeviceSynchronize());
            const double d2 = now() - t1;
            printf("2D pitched %s: %6.1f GB/s each way\n", mode == 0 ? "H2D alone " : mode == 1 ? "D2H alone " : "both      ", (double)wb * rows * pages * 4 / 1e9 / d2);
        }
    }
    return 0;
}
