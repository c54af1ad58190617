// hOCR-box mask commit (reference: mrc.py:265-266 `mask_arr[top:bottom, left:right] = th`
// executed box after box).  Later boxes overwrite earlier ones where they overlap, so a
// pixel of box b is written only if no later box with a decision covers it.  One launch
// for all boxes of a page: grid = (column tiles, rows, boxes of all pages of the batch).
#include "mrchip_internal.h"

namespace mrchip {

__global__ __launch_bounds__(256) void hocr_commit_kernel(const HocrBox *boxes) {
    const int b = blockIdx.z;
    const HocrBox B = boxes[b];
    uint8_t *mask = B.mask;
    const int mpitch = B.mpitch, nb = B.page_end;
    if (B.decision == 0) return;
    const int bw = B.r - B.l, bh = B.b - B.t;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= bw) return;
    const uint8_t *th = B.decision == 1 ? B.th : B.thi;
    for (int y = blockIdx.y; y < bh; y += gridDim.y) {
        const int px = B.l + x, py = B.t + y;
        bool covered = false;
        for (int j = b + 1; j < nb && !covered; j++) {
            const HocrBox &L = boxes[j];
            covered = L.decision != 0 && px >= L.l && px < L.r && py >= L.t && py < L.b;
        }
        if (!covered) mask[(size_t)py * mpitch + px] = th[(size_t)y * B.pitch + x];
    }
}

int launch_hocr_commit(mrchip_ctx *ctx, hipStream_t s, const HocrBox *d_boxes, int nb, int maxw, int maxh, double area) {
    if (nb <= 0) return 0;
    dim3 grid(cdiv(maxw, 256), std::min(maxh, 64), nb);
    LAUNCH(ctx, s, "hocr_commit", 2.0 * area,
           hipLaunchKernelGGL(hocr_commit_kernel, grid, dim3(256), 0, s, d_boxes));
    return 0;
}

}  // namespace mrchip
