// micro-benchmark: issue cost of the VALU / LDS instructions the Sauvola and optimise kernels are built from,
// in cycles per wave64 instruction per SIMD with 8 waves per SIMD resident (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
    unsigned a[8], b[8];
    double d[8];
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * seed;
    __syncthreads();
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i; b[i] = seed + 17 * i + threadIdx.x; d[i] = 1.0 + a[i] * 1e-9; }
    const unsigned m = seed | 0x10001u;
    const float fm = 1.0000001f;
    const double dm = 1.00000000001;
    const unsigned la = (threadIdx.x * 8) & 0x3ff8;
    for (int it = 0; it < N_ITER; it++) {
#define X(i) \
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 1) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); \
        if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm)); \
        if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(dm)); \
        if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(dm)); \
        if (OP == 9) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i])); \
        if (OP == 10) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 11) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i])); \
        if (OP == 12) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 14) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 15) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm)); \
        if (OP == 16) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm)); \
        if (OP == 17) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm)); \
        if (OP == 18) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[i]) : "v"(a[i])); \
        if (OP == 19) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i])); \
        if (OP == 20) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i])); \
        if (OP == 21) asm volatile("v_cmp_le_f64 vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %2, vcc" : : "v"(d[i]), "v"(dm), "v"(a[i]) : "vcc"); \
        if (OP == 22) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 23) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); \
        if (OP == 24) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 25) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
        if (OP == 26) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i])); \
        if (OP == 27) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 28) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i])); \
        if (OP == 29) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(fm) : "vcc"); \
        if (OP == 30) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); \
        if (OP == 31) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(m) : "vcc"); \
        if (OP == 32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[i]) : "v"(la), "n"(i * 256)); \
        if (OP == 33) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i]) : "v"(la), "n"(i * 256)); \
        if (OP == 34) asm volatile("ds_write_b32 %1, %0 offset:%2" : : "v"(b[i]), "v"(la), "n"(i * 256)); \
        if (OP == 35) asm volatile("ds_write_b64 %1, %0 offset:%2" : : "v"(d[i]), "v"(la), "n"(i * 256)); \
        if (OP == 36) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(a[i])); \
        if (OP == 37) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 38) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2" : "+v"(a[i]) : "v"(m)); \
        if (OP == 39) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); \
        if (OP == 40) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i])); \
        if (OP == 41) asm volatile("v_cvt_u32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i])); \
        if (OP == 42) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i])); \
        if (OP == 43) asm volatile("v_cvt_pk_u8_f32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 44) asm volatile("v_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); \
        if (OP == 45) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m)); \
        if (OP == 46) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 48) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 49) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 50) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i])); \
        if (OP == 51) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i])); \
        if (OP == 52) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 53) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 54) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc"); \
        if (OP == 55) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(fm) : "vcc"); \
        if (OP == 56) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i])); \
        if (OP == 57) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i])); \
        if (OP == 58) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 59) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 60) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 61) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(fm), "v"(b[i])); \
        if (OP == 62) asm volatile("v_fma_f32 %0, |%0|, %1, -%0" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 63) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i])); \
        if (OP == 64) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc"); \
        if (OP == 65) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(m) : "vcc"); \
        if (OP == 66) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 67) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 68) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fm)); \
        if (OP == 69) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i])); \
        if (OP == 70) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 71) asm volatile("v_add_u32 %0, %0, %2\n v_mad_u32_u24 %1, %1, %2, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(m)); \
        if (OP == 72) asm volatile("v_fma_f32 %0, %0, %2, %0\n v_fma_f64 %1, %1, %3, %1" : "+v"(a[i]), "+v"(d[i]) : "v"(fm), "v"(dm)); \
        if (OP == 73) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 74) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(m)); \
        if (OP == 75) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(d[i]), "v"(dm) : "vcc"); \
        if (OP == 76) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(seed)); \
        if (OP == 77) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(seed)); \
        if (OP == 78) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); \
        if (OP == 47) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(m));
        R8(X)
#undef X
        if (OP >= 32 && OP <= 35) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r += a[i] + b[i] + (unsigned)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(const char *name) {
    unsigned *d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;     // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 3u);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a); k<OP><<<blocks, 256>>>(d, 3u); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    double winstr = (double)blocks * 4 / 1024 * N_ITER * 8;
    printf("%-34s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, best, best * 1e-3 * 2.4e9 / winstr);
    hipFree(d);
}
int main() {
    run<0>("v_add_u32"); run<1>("v_mad_u32_u24"); run<39>("v_mad_i32_i24"); run<2>("v_mul_lo_u32"); run<3>("v_mul_hi_u32"); run<4>("v_mul_hi_u32_u24");
    run<31>("v_mad_u64_u32"); run<30>("v_dot4_u32_u8"); run<45>("v_sad_u8"); run<44>("v_mad_u16");
    run<22>("v_pk_add_u16"); run<23>("v_pk_mad_u16"); run<46>("v_pk_mul_lo_u16");
    run<24>("v_add_u32_sdwa"); run<38>("v_sub_u32_sdwa(b,b)"); run<25>("v_add_u32_dpp row_shr"); run<26>("v_perm_b32"); run<27>("v_lshl_or_b32"); run<28>("v_bfe_u32"); run<47>("v_alignbit_b32");
    run<37>("v_mul_f32"); run<5>("v_fma_f32"); run<6>("v_pk_fma_f32"); run<7>("v_pk_mul_f32"); run<8>("v_pk_add_f32");
    run<9>("v_cvt_f32_u32"); run<10>("v_cvt_u32_f32"); run<11>("v_cvt_f32_ubyte1"); run<43>("v_cvt_pk_u8_f32"); run<12>("v_floor_f32"); run<13>("v_rcp_f32"); run<14>("v_sqrt_f32"); run<42>("v_rsq_f32");
    run<29>("v_cmp_le_f32 + v_cndmask (2 instr)");
    run<15>("v_fma_f64"); run<16>("v_mul_f64"); run<17>("v_add_f64"); run<18>("v_cvt_f64_u32"); run<36>("v_cvt_f64_f32"); run<19>("v_floor_f64"); run<20>("v_cvt_f32_f64"); run<41>("v_cvt_u32_f64"); run<40>("v_rcp_f64");
    run<21>("v_cmp_le_f64 + v_addc (2 instr)");
    run<48>("v_and_b32"); run<49>("v_or_b32"); run<66>("v_xor_b32"); run<50>("v_lshlrev_b32"); run<51>("v_lshrrev_b32"); run<52>("v_sub_u32"); run<78>("v_subrev_u32"); run<70>("v_min_u32");
    run<53>("v_add_f32"); run<68>("v_sub_f32"); run<58>("v_max_f32"); run<67>("v_mul_f32 |abs|"); run<61>("v_fmac_f32"); run<62>("v_fma_f32 abs/neg"); run<77>("v_fma_f32 sgpr operand");
    run<54>("v_cndmask_b32"); run<55>("v_cmp_le_f32"); run<64>("v_cmp_le_u32"); run<75>("v_cmp_le_f64"); run<56>("v_mov_b32"); run<57>("v_add3_u32"); run<63>("v_and_or_b32"); run<73>("v_lshl_add_u32"); run<74>("v_add_lshl_u32"); run<65>("v_add_co_u32");
    run<59>("v_mul_u32_u24"); run<60>("v_mul_i32_i24"); run<76>("v_mad_u32_u24 sgpr operand"); run<69>("v_cvt_f32_i32");
    run<71>("v_add_u32 + v_mad_u32_u24 (2 instr)"); run<72>("v_fma_f32 + v_fma_f64 (2 instr)");
    run<32>("ds_read_b32 (8 per wait)"); run<33>("ds_read_b64 (8 per wait)"); run<34>("ds_write_b32"); run<35>("ds_write_b64");
    return 0;
}
