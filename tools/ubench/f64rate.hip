// micro-benchmark: issue rate of fp64 / conversion instructions on gfx950 (cycles per wave64 instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a0, unsigned u0) {
    double x0 = a0 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u = u0 + threadIdx.x;
    float f0 = (float)x0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    for (int i = 0; i < N_ITER; i++) {
        if (OP == 0) { x0 = __dmul_rn(x0, a0); x1 = __dmul_rn(x1, a0); x2 = __dmul_rn(x2, a0); x3 = __dmul_rn(x3, a0); x4 = __dmul_rn(x4, a0); x5 = __dmul_rn(x5, a0); x6 = __dmul_rn(x6, a0); x7 = __dmul_rn(x7, a0); }
        if (OP == 1) { x0 = __dadd_rn(x0, a0); x1 = __dadd_rn(x1, a0); x2 = __dadd_rn(x2, a0); x3 = __dadd_rn(x3, a0); x4 = __dadd_rn(x4, a0); x5 = __dadd_rn(x5, a0); x6 = __dadd_rn(x6, a0); x7 = __dadd_rn(x7, a0); }
        if (OP == 2) { x0 = __fma_rn(x0, a0, a0); x1 = __fma_rn(x1, a0, a0); x2 = __fma_rn(x2, a0, a0); x3 = __fma_rn(x3, a0, a0); x4 = __fma_rn(x4, a0, a0); x5 = __fma_rn(x5, a0, a0); x6 = __fma_rn(x6, a0, a0); x7 = __fma_rn(x7, a0, a0); }
        if (OP == 3) { x0 += (double)(u + i); x1 += (double)(u ^ i); x2 += (double)(u + 2 * i); x3 += (double)(u + 3 * i); x4 += (double)(u + 5 * i); x5 += (double)(u + 7 * i); x6 += (double)(u + 9 * i); x7 += (double)(u + 11 * i); }
        if (OP == 4) { f0 = __fmul_rn(f0, (float)a0); f1 = __fmul_rn(f1, (float)a0); f2 = __fmul_rn(f2, (float)a0); f3 = __fmul_rn(f3, (float)a0); f4 = __fmul_rn(f4, (float)a0); f5 = __fmul_rn(f5, (float)a0); f6 = __fmul_rn(f6, (float)a0); f7 = __fmul_rn(f7, (float)a0); }
        if (OP == 5) { u = u * 2654435761u + i; u = u * 2246822519u + 1; u = u * 3266489917u + 2; u = u * 668265263u + 3; u = u * 374761393u + 4; u = u * 2654435761u + 5; u = u * 2246822519u + 6; u = u * 3266489917u + 7; }
        if (OP == 7) { x0 = __builtin_floor(x0) + a0; x1 = __builtin_floor(x1) + a0; x2 = __builtin_floor(x2) + a0; x3 = __builtin_floor(x3) + a0; x4 = __builtin_floor(x4) + a0; x5 = __builtin_floor(x5) + a0; x6 = __builtin_floor(x6) + a0; x7 = __builtin_floor(x7) + a0; }
        if (OP == 8) { u += (unsigned)x0; x0 += 1.0; u += (unsigned)x1; x1 += 1.0; u += (unsigned)x2; x2 += 1.0; u += (unsigned)x3; x3 += 1.0; u += (unsigned)x4; x4 += 1.0; u += (unsigned)x5; x5 += 1.0; u += (unsigned)x6; x6 += 1.0; u += (unsigned)x7; x7 += 1.0; }
        if (OP == 9) { u += (x0 <= x1) ? 1u : 0u; x0 += a0; u += (x2 <= x3) ? 1u : 0u; x2 += a0; u += (x4 <= x5) ? 1u : 0u; x4 += a0; u += (x6 <= x7) ? 1u : 0u; x6 += a0; u += (x1 <= x2) ? 1u : 0u; x1 += a0; u += (x3 <= x4) ? 1u : 0u; x3 += a0; u += (x5 <= x6) ? 1u : 0u; x5 += a0; u += (x7 <= x0) ? 1u : 0u; x7 += a0; }
        if (OP == 10) { x0 += (double)(float)(u + i); x1 += (double)(float)(u ^ i); x2 += (double)(float)(u + 2 * i); x3 += (double)(float)(u + 3 * i); x4 += (double)(float)(u + 5 * i); x5 += (double)(float)(u + 7 * i); x6 += (double)(float)(u + 9 * i); x7 += (double)(float)(u + 11 * i); }
        if (OP == 6) { x0 = (double)(float)x0; x1 = (double)(float)x1; x2 = (double)(float)x2; x3 = (double)(float)x3; x4 = (double)(float)x4; x5 = (double)(float)x5; x6 = (double)(float)x6; x7 = (double)(float)x7; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + u + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int OP>
double run(const char *name, double ops_per_iter) {
    double *d; hipMalloc(&d, 1024 * 256 * 8 * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;     // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 1.0000001, 3u);
    hipEventRecord(a); k<OP><<<blocks, 256>>>(d, 1.0000001, 3u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * N_ITER * ops
    double winstr = (double)blocks * 4 / 1024 * N_ITER * ops_per_iter;
    double cyc = ms * 1e-3 * 2.4e9 / winstr;
    printf("%-28s %.3f ms  ~%.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc);
    hipFree(d); return cyc;
}
int main() {
    run<4>("v_mul_f32", 8); run<0>("v_mul_f64", 8); run<1>("v_add_f64", 8); run<2>("v_fma_f64", 8);
    run<3>("v_cvt_f64_u32 + v_add_f64", 16); run<7>("v_floor_f64 + v_add_f64", 16); run<8>("v_cvt_u32_f64 + add_f64 + add_u32", 24);
    run<9>("v_cmp_le_f64 + cndmask/add + add_f64", 24); run<10>("cvt_f32_u32 + cvt_f64_f32 + add_f64", 24); run<6>("cvt f64->f32->f64", 16); run<5>("v_mul_lo_u32 (+add)", 16);
    return 0;
}
