// hOCR-box mask commit (reference: mrc.py:265-266 `mask_arr[top:bottom, left:right] = th`
// executed box after box).  Later boxes overwrite earlier ones where they overlap, so a
// pixel of box b is written only if no later box with a decision covers it.  One launch
// for all boxes of a page: grid = (column tiles, rows, boxes of all pages of the batch).
#include "mrchip_internal.h"

namespace mrchip {

// four 0/1 bytes -> four bits (byte k -> bit k)
__device__ __forceinline__ unsigned nib01(unsigned x) { return ((x * 0x01020408u) >> 24) & 0xFu; }

__global__ __launch_bounds__(256) void hocr_commit_kernel(const HocrBox *boxes, int first, int or_mode) {
    const int b = first + blockIdx.z;
    const HocrBox B = boxes[b];
    if (B.decision == 0) return;
    uint8_t *mask = B.mask;
    const int mpitch = B.mpitch, nb = B.page_end;
    const int bh = B.b - B.t;
    // one lane per 16 pixels, 16-byte aligned in the MASK's coordinates (the box scratch has the same column phase
    // mod 16, so its 16-byte groups line up too): lanes wholly inside the box move one uint4 each way
    const int xa0 = (B.l & ~15) + (blockIdx.x * 256 + threadIdx.x) * 16;   // absolute column of the lane's first pixel
    if (xa0 >= B.r) return;
    const uint8_t *th = B.decision == 1 ? B.th : B.thi;
    const bool whole = xa0 >= B.l && xa0 + 16 <= B.r;
    // global (not flat) accesses on the main path: the pointers come out of a record, so hipcc would emit flat_* (which
    // also count on lgkmcnt and are the slower instruction)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(1))) *gc_u4p;
    typedef u32x4 __attribute__((address_space(1))) *g_u4p;
    typedef unsigned short __attribute__((address_space(1))) *g_u16p;
    for (int y = blockIdx.y; y < bh; y += gridDim.y) {
        const int py = B.t + y;
        const uint8_t *tp = th + (ptrdiff_t)y * B.pitch + (xa0 - B.l);
        uint8_t *mp = mask + (size_t)py * mpitch + xa0;
        if (whole && !B.overlapped) {
            // or_mode: the mask already holds the page threshold (mrc.py:329's OR, applied first); each pixel has
            // exactly one owner (box, lane), so the read-modify-write is race-free
            u32x4 v = *(gc_u4p)(uintptr_t)tp;
            if (B.bits && B.no_bytes) {
                // only the 1-bpp rows carry the mask at this point (the bytes are rewritten from them after the denoiser):
                // the lane owns these 16 bits, which hold the page threshold -- OR the box threshold in
                const g_u16p bp = (g_u16p)(uintptr_t)(B.bits + (size_t)py * B.bits_pitch + (xa0 >> 3));
                *bp = (unsigned short)(*bp | nib01(v.x) | nib01(v.y) << 4 | nib01(v.z) << 8 | nib01(v.w) << 12);
                continue;
            }
            if (or_mode) v |= *(gc_u4p)(uintptr_t)mp;
            *(g_u4p)(uintptr_t)mp = v;
            if (B.bits) {
                // the same 16 pixels of the 1-bpp row: the lane owns the whole group and holds its final bytes (page
                // threshold | box threshold), so it STORES the 16 bits -- no read-modify-write, no atomic
                const unsigned m16 = nib01(v.x) | nib01(v.y) << 4 | nib01(v.z) << 8 | nib01(v.w) << 12;
                *(g_u16p)(uintptr_t)(B.bits + (size_t)py * B.bits_pitch + (xa0 >> 3)) = (unsigned short)m16;
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int xa = xa0 + 4 * q;
            if (xa >= B.r) break;
            if (xa + 4 <= B.l) continue;
            unsigned keep = 0;                       // byte mask of pixels this box owns
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int px = xa + i;
                if (px >= B.l && px < B.r) keep |= 0xffu << (8 * i);
            }
            if (B.overlapped) {                      // rare: some later box with a decision intersects this one
                for (int j = b + 1; j < nb; j++) {
                    const HocrBox &L = boxes[j];
                    if (L.decision == 0 || py < L.t || py >= L.b) continue;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int px = xa + i;
                        if (px >= L.l && px < L.r) keep &= ~(0xffu << (8 * i));
                    }
                }
            }
            if (!keep) continue;
            if (B.bits) {         // partial group: another box may own other bits of the word -> atomic OR of the owned ones
                unsigned tv = 0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if ((keep >> (8 * i)) & 0xffu) tv |= (unsigned)(tp[4 * q + i] & 1u) << i;
                if (tv) atomicOr(reinterpret_cast<unsigned *>(B.bits + (size_t)py * B.bits_pitch) + (xa >> 5), tv << (xa & 31));
            }
            if (B.bits && B.no_bytes) continue;       // (the bit rows have it: see above)
            uint8_t *mq8 = mp + 4 * q;
            if (keep == 0xffffffffu) {
                const unsigned v = *reinterpret_cast<const unsigned *>(tp + 4 * q);
                unsigned *mq = reinterpret_cast<unsigned *>(mq8);
                *mq = or_mode ? (*mq | v) : v;
            } else {      // partial dword: byte accesses, so that a neighbouring box's bytes are never touched (and the
                          // scratch is never read in front of the box's first column)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if ((keep >> (8 * i)) & 0xffu) {
                        const uint8_t bv = tp[4 * q + i];
                        mq8[i] = or_mode ? (uint8_t)(mq8[i] | bv) : bv;
                    }
            }
        }
    }
}

int launch_hocr_commit(mrchip_ctx *ctx, hipStream_t s, const HocrBox *d_boxes, int nb, int maxw, int maxh, double area,
                       int or_mode) {
    if (nb <= 0) return 0;
    // grid.z is limited to 65535: big batches of small pages can hold more boxes than that
    for (int first = 0; first < nb; first += MAX_GRID_Z) {
        const int cnt = std::min(MAX_GRID_Z, nb - first);
        dim3 grid(cdiv(cdiv(maxw + 15, 16) + 1, 256), std::min(maxh, 64), cnt);
        LAUNCH(ctx, s, "hocr_commit", (or_mode ? 3.0 : 2.0) * area * cnt / nb,
               hipLaunchKernelGGL(hocr_commit_kernel, grid, dim3(256), 0, s, d_boxes, first, or_mode));
    }
    return 0;
}

}  // namespace mrchip
