// probe: rounding / saturation of v_cvt_pk_u8_f32 on gfx950 (the optimise kernels pack their quotient bytes with it)
//   hipcc --offload-arch=gfx950 -O3 -o cvt_pk_u8 cvt_pk_u8.hip && ./cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        unsigned r = 0xAABBCCDDu;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r) : "v"(in[i]));
        out[i] = r;
    }
}
int main() {
    float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 2.51f, 3.5f, 254.5f, 254.99f, 255.0f, 255.5f, 256.0f, 300.0f, -0.4f,
                 -0.6f, -1.0f, -5.0f, NAN, INFINITY, -INFINITY, 1e30f};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d; unsigned *o, ho[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12g -> byte1 = %3u   (dword %08x)\n", h[i], (ho[i] >> 8) & 0xff, ho[i]);
    return 0;
}
