// Measurement tool, not part of the library: what a plain streaming kernel reaches on THIS box -- copy (read + write), read only, write only --
// to read bench.py's `hbm_copy_GBps_measured` (hipMemcpyAsync device-to-device, the runtime's blit) and the roofline fractions against.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/hbm_copy_peak.hip -o tools/ubench/bin/hbm_copy_peak && tools/ubench/bin/hbm_copy_peak [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}

__global__ __launch_bounds__(256) void read_kernel(const f4 *__restrict__ a, float *out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;        // never true: keeps the loads
}

__global__ __launch_bounds__(256) void write_kernel(f4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = v;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 1.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)4095, n = bytes / 16;
    f4 *a, *b; float *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    auto timed = [&](const char *name, double moved, auto launch) -> int {
        launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f GB/s\n", name, moved * reps / (ms * 1e-3) / 1e9);
        return 0;
    };
    char name[96];
    if (timed("hipMemcpyAsync device to device (r + w)", 2.0 * bytes, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); })) return 1;
    for (int wgs : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
        snprintf(name, sizeof name, "float4 copy, %d workgroups (r + w)", wgs);
        if (timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(wgs), dim3(256), 0, 0, a, b, n); })) return 1;
    }
    for (int wgs : {256 * 8, 256 * 32}) {
        snprintf(name, sizeof name, "float4 copy non-temporal, %d workgroups", wgs);
        if (timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(wgs), dim3(256), 0, 0, a, b, n); })) return 1;
    }
    for (int wgs : {256 * 8, 256 * 32}) {
        snprintf(name, sizeof name, "float4 read only, %d workgroups", wgs);
        if (timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL(read_kernel, dim3(wgs), dim3(256), 0, 0, a, o, n); })) return 1;
        snprintf(name, sizeof name, "float4 write only, %d workgroups", wgs);
        if (timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL(write_kernel, dim3(wgs), dim3(256), 0, 0, b, n); })) return 1;
    }
    // a copy with one-element-per-thread grid (no loop): the shape of luma601 / unpack
    return 0;
}
