// Probe of v_mfma_i32_16x16x64_i8 operand slots on gfx950: which (lane group, byte) slot of A is
// multiplied with which slot of B, and where D[m][n] lands.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(int *pair, int *dmap) {
    const int lane = threadIdx.x;
    // pairing: A(row 0, slot s) = 1; B(col 0, slot t) = t + 1
    for (int s = 0; s < 64; s++) {
        v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0};
        if ((lane & 15) == 0) {
            const int g = lane >> 4;
            unsigned bb[4] = {0, 0, 0, 0};
            for (int i = 0; i < 16; i++) bb[i >> 2] |= (unsigned)(g * 16 + i + 1) << (8 * (i & 3));
            b = (v4i){(int)bb[0], (int)bb[1], (int)bb[2], (int)bb[3]};
            if (g == (s >> 4)) { unsigned aa[4] = {0, 0, 0, 0}; aa[(s & 15) >> 2] = 1u << (8 * (s & 3)); a = (v4i){(int)aa[0], (int)aa[1], (int)aa[2], (int)aa[3]}; }
        }
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
        if (lane == 0) pair[s] = c[0] - 1;
    }
    // D map: A(row m, slot 0) = m + 1, B(col n, slot 0) = n + 1 (lanes 0..15 only) -> D[m][n] = (m+1)(n+1)
    {
        v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0};
        if (lane < 16) { a[0] = lane + 1; b[0] = 2 * lane + 3; }
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
        for (int i = 0; i < 4; i++) dmap[lane * 4 + i] = c[i];
    }
}
__global__ void signs(int *out) {
    const int lane = threadIdx.x;
    v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0};
    if (lane == 0) { a[0] = 0xCD; b[0] = 3; }            // A = -51 (or 205), B = 3
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
    a = (v4i){0, 0, 0, 0}; b = a; c = a;
    if (lane == 0) { a[0] = 3; b[0] = 0xCD; }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    if (lane == 0) out[1] = c[0];
    // all 16 bytes of lane 0: A = byte index + 1, B = 1 -> 136; A at byte i only times B = i + 1
    a = (v4i){0x04030201, 0x08070605, 0x0c0b0a09, 0x100f0e0d}; b = (v4i){0x01010101, 0x01010101, 0x01010101, 0x01010101}; c = (v4i){0, 0, 0, 0};
    if (lane != 0) { a = (v4i){0, 0, 0, 0}; b = a; }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    if (lane == 0) out[2] = c[0];
}
int main() {
    { int *o; hipMalloc(&o, 16); signs<<<1, 64>>>(o); int h[3]; hipMemcpy(h, o, 12, hipMemcpyDeviceToHost);
      printf("A=0xCD*B=3 -> %d ; A=3*B=0xCD -> %d (signed: -153) ; sum 1..16 -> %d (136)\n", h[0], h[1], h[2]); }
    int *pair, *dmap;
    hipMalloc(&pair, 64 * 4); hipMalloc(&dmap, 256 * 4);
    probe<<<1, 64>>>(pair, dmap);
    int hp[64], hd[256];
    hipMemcpy(hp, pair, sizeof hp, hipMemcpyDeviceToHost); hipMemcpy(hd, dmap, sizeof hd, hipMemcpyDeviceToHost);
    printf("A slot -> paired B slot:"); for (int s = 0; s < 64; s++) printf(" %d", hp[s]); printf("\n");
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
        const int m = (l >> 4) * 4 + i, n = l & 15;
        if (hd[l * 4 + i] != (m + 1) * (2 * n + 3)) bad++;
    }
    printf("D[m=(l>>4)*4+i][n=l&15] with a=A(m), b=B(n): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) { for (int l = 0; l < 64; l += 5) printf("lane %d: %d %d %d %d\n", l, hd[l*4], hd[l*4+1], hd[l*4+2], hd[l*4+3]); }
    return 0;
}
