// Device-resident page batches: the three yields of mrc.create_mrc_hocr_components
// (reference: internetarchivepdf/mrc.py:334-471) for N same-sized pages at once, every
// stage one launch over the whole batch (grid.z / job arrays), pixels crossing PCIe once.
//
//   mask_begin  (enqueue)  luma (mrc.py:361); both hOCR-box thresholds + set-pixel counts
//                          (mrc.py:223-238); noise estimate of the central crop (mrc.py:305)
//   sigmas      (sync)     sigma_est per page back to the host, which -- like scipy -- owns the
//                          Gaussian weight tables
//   mask_finish (enqueue)  box decisions (mrc.py:240-263) + ordered commit (mrc.py:266);
//                          blur (mrc.py:309-311); Sauvola k=0.34 OR-ed into the mask
//                          (mrc.py:325-329); fast denoise (mrc.py:388)
//   layers      (enqueue)  optimise fg (n=3) / bg (n=10, inverted mask) (mrc.py:412-415,
//                          446-449) + thumbnail (mrc.py:420-434, 454-468)
// The single-page handle (mrchip_page_*) is a batch of one.
#include <algorithm>
#include <cmath>

#include "mrchip_internal.h"

using namespace mrchip;

struct BoxInfo {
    int page;
    int l, t, r, b;
    size_t off;       // byte offset of the box's scratch rows
    int pitch;
    int phase;
    int decision;
};

struct PlaneBuf {
    DevBuf buf;
    Plane pl;
    int alloc(mrchip_ctx *ctx, int npages, int row_bytes, int h, bool pitched = true) {
        pl.pitch = pitched ? round_up(row_bytes + 64, 64) : row_bytes;
        pl.stride = ((size_t)pl.pitch * h + 2 * PAD + 255) & ~(size_t)255;
        TRY(buf.alloc(ctx, pl.stride * npages + 4096));
        pl.p = buf.as<uint8_t>() + PAD;
        return 0;
    }
};

struct mrchip_batch {
    mrchip_ctx *ctx = nullptr;
    hipStream_t s = nullptr;
    int n = 0, w = 0, h = 0, c = 1;
    int active = 0;                  // pages in use (<= n): what every stage launch covers (mrchip_batch_set_count)
    PlaneBuf img, gray_own, blur, mask, layer[2], small[2], sc1[2], sc2[2];
    Plane gray;
    DevBuf gtmp;  int gtmp_pitch = 0;  size_t gtmp_stride = 0;     // float32 scratch of the blur
    DevBuf sig_scratch;  size_t sig_stride = 0;
    DevBuf box_sig_scratch, dn_bits, ctrl, thA, thB, tables[2];
    OptMail opt_mail;                 // hand-off granules of optimise's column-strip schedule
    std::vector<int> need;  std::vector<double> ratio, inv_ratio;      // box decisions in flight (mask_finish)
    hipEvent_t box_ev = nullptr;  size_t box_sig_cap = 0;
    int bits_valid = 0;                       // dn_bits holds the finished masks at 1 bpp (fast denoise ran)
    bool commit_bits = false;                 // mask_finish in flight with the 1-bpp rows written by Sauvola + commit
    DevBuf packed;  int packed_valid = 0;     // 1-bpp copy of the finished masks (made on first request)
    size_t dn_stride = 0, th_bytes = 0;
    ThumbPlan plan[2];
    int plan_req[2][2] = {{0, 0}, {0, 0}};
    int layer_w[2] = {0, 0}, layer_h[2] = {0, 0}, layer_small[2] = {0, 0}, layer_done[2] = {0, 0};
    int crop[4] = {0, 0, 0, 0};      // hs, he, ws, we of mrc.estimate_noise
    // pinned host mirror of the control block
    unsigned char *hctrl = nullptr;
    size_t ctrl_bytes = 0;
    std::vector<std::vector<int32_t>> page_boxes;   // per page, 4 ints per box
    std::vector<BoxInfo> boxes;                     // all pages, page-major, list order
    std::vector<int> first_box;                     // per page: index of its first box (+ sentinel)
    std::vector<double> sigma;
    int window = 51;
    int state = 0;    // 0 created, 1 uploaded, 2 mask_begin, 3 sigma known, 4 mask done
    // The control block is laid out for nb_cap boxes, a capacity that only grows (with a stream
    // synchronisation): the offsets of its regions then stay put from step to step, so a region is only
    // rewritten after the host has waited for the stream at least once since its last copy was queued
    // (mrchip_batch_sigmas sits between any two writes of the same region).
    int nb_cap = 0;
    std::vector<char> gray_given;     // per page: the caller supplied the gray plane (mrchip_batch_upload_gray)
};

// control block layout (device + pinned mirror)
struct CtrlLayout {
    size_t sigma, box_sigma, counts, jobs, pjobs, boxes, gauss, optjobs, sigjobs, boxsigjobs, total;
    CtrlLayout(int npages, int nb) {
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o = (o + bytes + 63) & ~(size_t)63; return r; };
        sigma = take((size_t)npages * 8);
        box_sigma = take((size_t)nb * 16 + 16);
        counts = take((size_t)nb * 8);
        jobs = take((size_t)nb * sizeof(SauvolaJob));
        pjobs = take((size_t)npages * sizeof(SauvolaJob));
        boxes = take((size_t)nb * sizeof(HocrBox));
        gauss = take((size_t)npages * sizeof(GaussW));
        optjobs = take((size_t)npages * 2 * sizeof(OptJob));
        sigjobs = take((size_t)npages * sizeof(SigJob));
        boxsigjobs = take((size_t)nb * 2 * sizeof(SigJob));
        total = o + 64;
    }
};

#define CHECK_B(b)                                                          \
    do {                                                                    \
        if (!(b)) { set_error("null batch/page"); return MRCHIP_E_ARG; }    \
        HIP_TRY(hipSetDevice((b)->ctx->device));                            \
    } while (0)

static int batch_init(mrchip_batch *b, mrchip_ctx *ctx, int npages, int w, int h, int c) {
    b->ctx = ctx;
    b->s = ctx->streams[ctx->next_stream++ % NSTREAMS];
    b->n = npages; b->active = npages; b->w = w; b->h = h; b->c = c;
    TRY(b->img.alloc(ctx, npages, w * c, h));
    if (c == 3) TRY(b->gray_own.alloc(ctx, npages, w, h));
    TRY(b->mask.alloc(ctx, npages, w, h));
    TRY(b->blur.alloc(ctx, npages, w, h));
    b->gray = c == 3 ? b->gray_own.pl : b->img.pl;
    b->gtmp_pitch = round_up(w, 16);
    b->gtmp_stride = (size_t)b->gtmp_pitch * h;
    TRY(b->gtmp.alloc(ctx, b->gtmp_stride * sizeof(float) * npages));
    int hs = (int)(h / 2.0 - h / 4.0), he = (int)(h / 2.0 + h / 4.0);      // mrc.py:282-285
    int ws = (int)(w / 2.0 - w / 4.0), we = (int)(w / 2.0 + w / 4.0);
    if (he == 0 || we == 0) { hs = 0; he = h; ws = 0; we = w; }           // mrc.py:288-292
    b->crop[0] = hs; b->crop[1] = he; b->crop[2] = ws; b->crop[3] = we;
    b->sig_stride = sigma_scratch_bytes(we - ws, he - hs, 0);
    TRY(b->sig_scratch.alloc(ctx, b->sig_stride * npages));
    b->dn_stride = (denoise_scratch_bytes(w, h) + 3) / 4;
    TRY(b->dn_bits.alloc(ctx, b->dn_stride * 4 * npages));
    // the 1-bpp rows' tails (bits past the last column inside a row's last word) must be zero and the fused producers
    // (Sauvola, hOCR commit) never write there
    HIP_TRY(hipMemsetAsync(b->dn_bits.p, 0, b->dn_stride * 4 * npages, b->s));
    b->page_boxes.resize(npages);
    b->gray_given.assign(npages, 0);
    b->sigma.assign(npages, 0.0);
    return 0;
}

MRCHIP_EXPORT mrchip_batch *mrchip_batch_create(mrchip_ctx *ctx, int npages, int w, int h, int channels) {
    if (!ctx || npages <= 0 || w <= 0 || h <= 0 || (channels != 1 && channels != 3)) {
        set_error("batch_create: bad arguments (npages=%d w=%d h=%d channels=%d)", npages, w, h, channels);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    mrchip_batch *b = new mrchip_batch();
    if (batch_init(b, ctx, npages, w, h, channels)) { delete b; return nullptr; }
    return b;
}

MRCHIP_EXPORT void mrchip_batch_destroy(mrchip_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->s);
    if (b->hctrl) (void)hipHostFree(b->hctrl);
    if (b->box_ev) (void)hipEventDestroy(b->box_ev);
    delete b;
}

// new pixels or boxes: everything derived from the previous ones is stale (mask, 1-bpp rows, layers)
static void invalidate_results(mrchip_batch *b) {
    b->state = 1;
    b->bits_valid = 0;
    b->packed_valid = 0;
    b->layer_done[0] = b->layer_done[1] = 0;
}

MRCHIP_EXPORT int mrchip_batch_upload(mrchip_batch *b, int page, const uint8_t *img) {
    CHECK_B(b);
    if (!img || page < 0 || page >= b->n) { set_error("batch_upload: bad arguments"); return MRCHIP_E_ARG; }
    TRY(upload_2d(b->s, b->img.pl.page(page), b->img.pl.pitch, img, b->w * b->c, b->w * b->c, b->h));
    invalidate_results(b);
    b->gray_given[page] = 0;
    return 0;
}

// Gray plane of an RGB batch page computed by the caller: create_mrc_hocr_components takes
// `image.convert('L')` of the ORIGINAL image (mrc.py:359-361) and only later converts modes other than
// L / RGB to RGB for the layers (mrc.py:401-404); for those modes Pillow's L conversion is not the luma of
// the RGB conversion, so the mirror uploads both.  Call after mrchip_batch_upload of the same page.
MRCHIP_EXPORT int mrchip_batch_upload_gray(mrchip_batch *b, int page, const uint8_t *gray) {
    CHECK_B(b);
    if (!gray || page < 0 || page >= b->n) { set_error("batch_upload_gray: bad arguments"); return MRCHIP_E_ARG; }
    if (b->c != 3) { set_error("batch_upload_gray: the batch is gray already (upload the page itself)"); return MRCHIP_E_ARG; }
    TRY(upload_2d(b->s, b->gray_own.pl.page(page), b->gray_own.pl.pitch, gray, b->w, b->w, b->h));
    invalidate_results(b);
    b->gray_given[page] = 1;
    return 0;
}

// Replace the finished mask of a page (uint8/bool[h][w]) before the layers are made: the hook for mask
// post-processing that stays on the host, i.e. denoise_mask='bregman' (mrc.py:391-392, scikit-image's
// iterative TV solver; SURVEY.md 8f rank 4 keeps it a CPU passthrough).
MRCHIP_EXPORT int mrchip_batch_upload_mask(mrchip_batch *b, int page, const uint8_t *mask) {
    CHECK_B(b);
    if (!mask || page < 0 || page >= b->n) { set_error("batch_upload_mask: bad arguments"); return MRCHIP_E_ARG; }
    if (b->state < 4) { set_error("upload_mask before mask_finish"); return MRCHIP_E_STATE; }
    TRY(upload_2d(b->s, b->mask.pl.page(page), b->mask.pl.pitch, mask, b->w, b->w, b->h));
    b->bits_valid = 0;          // the denoiser's 1-bpp rows no longer describe the mask
    b->packed_valid = 0;
    b->layer_done[0] = b->layer_done[1] = 0;
    return 0;
}

// Number of pages in use, 1..npages (default npages): a streaming caller reuses one batch object for a short
// last batch; pages >= count are neither read nor written by any stage.
MRCHIP_EXPORT int mrchip_batch_set_count(mrchip_batch *b, int count) {
    CHECK_B(b);
    if (count < 1 || count > b->n) { set_error("batch_set_count: %d is outside 1..%d", count, b->n); return MRCHIP_E_ARG; }
    if (count != b->active) { b->active = count; invalidate_results(b); }
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_set_boxes(mrchip_batch *b, int page, const int32_t *boxes, int nb) {
    CHECK_B(b);
    if (page < 0 || page >= b->n || nb < 0 || (nb > 0 && !boxes)) { set_error("batch_set_boxes: bad arguments"); return MRCHIP_E_ARG; }
    for (int i = 0; i < nb; i++) {
        const int l = boxes[4 * i], t = boxes[4 * i + 1], r = boxes[4 * i + 2], bt = boxes[4 * i + 3];
        if (l < 0 || t < 0 || r > b->w || bt > b->h || l >= r || t >= bt) {
            set_error("batch_set_boxes: box %d (%d,%d,%d,%d) is not inside the %dx%d page (the caller filters, mrc.py:212-221)",
                      i, l, t, r, bt, b->w, b->h);
            return MRCHIP_E_ARG;
        }
    }
    b->page_boxes[page].assign(boxes, boxes + (size_t)4 * nb);
    if (b->state > 1) invalidate_results(b);
    return 0;
}

static int ensure_ctrl(mrchip_batch *b, int nb) {
    if (nb > b->nb_cap || !b->hctrl) {
        HIP_TRY(hipStreamSynchronize(b->s));
        b->nb_cap = std::max(nb + nb / 2, 64);
        const size_t need = (CtrlLayout(b->n, b->nb_cap).total + 4095) & ~(size_t)4095;
        if (b->hctrl) { HIP_TRY(hipHostFree(b->hctrl)); b->hctrl = nullptr; }
        TRY(b->ctrl.alloc(b->ctx, need));
        HIP_TRY(hipHostMalloc((void **)&b->hctrl, need, hipHostMallocDefault));
        b->ctrl_bytes = need;
    }
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_mask_begin(mrchip_batch *b, int window) {
    CHECK_B(b);
    if (b->state < 1) { set_error("mask_begin before upload"); return MRCHIP_E_STATE; }
    if (window < 1) { set_error("mask_begin: bad window"); return MRCHIP_E_ARG; }
    mrchip_ctx *ctx = b->ctx;
    hipStream_t s = b->s;
    const int w = b->w, h = b->h, N = b->active;
    b->window = window;
    if (b->c == 3) {                                                                                     // mrc.py:361
        for (int i = 0; i < N;) {          // runs of pages that have no caller-supplied gray plane
            if (b->gray_given[i]) { i++; continue; }
            int e = i;
            while (e < N && !b->gray_given[e]) e++;
            Plane src = b->img.pl, dst = b->gray_own.pl;
            src.p = src.page(i); dst.p = dst.page(i);
            TRY(launch_luma601(ctx, s, src, dst, w, h, e - i));
            i = e;
        }
    }
    // mask_arr = zeros (mrc.py:367) is never materialised: the page threshold is stored first and the hOCR
    // boxes are OR-ed on top (mask_finish), which is the same set of pixels as commit-then-OR (mrc.py:266, 329)
    // ---- hOCR boxes of all pages: both polarities + counts ----
    b->boxes.clear();
    b->first_box.assign(N + 1, 0);
    size_t off = 0;
    for (int pg = 0; pg < N; pg++) {
        b->first_box[pg] = (int)b->boxes.size();
        const std::vector<int32_t> &pb = b->page_boxes[pg];
        for (size_t i = 0; i + 3 < pb.size(); i += 4) {
            BoxInfo bi;
            bi.page = pg; bi.l = pb[i]; bi.t = pb[i + 1]; bi.r = pb[i + 2]; bi.b = pb[i + 3];
            bi.phase = bi.l & 15;
            bi.pitch = round_up(bi.r - bi.l + bi.phase, 16) + 16;
            bi.off = off;
            bi.decision = 0;
            off += (size_t)bi.pitch * (bi.b - bi.t);
            b->boxes.push_back(bi);
        }
    }
    const int nb = (int)b->boxes.size();
    b->first_box[N] = nb;
    TRY(ensure_ctrl(b, nb));
    const CtrlLayout L(b->n, b->nb_cap);
    unsigned char *dctrl = b->ctrl.as<unsigned char>();
    if (nb > 0) {
        if (off + 4096 > b->th_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            b->th_bytes = off + 4096 + off / 4;
            TRY(b->thA.alloc(ctx, b->th_bytes));
            TRY(b->thB.alloc(ctx, b->th_bytes));
        }
        SauvolaJob *hj = reinterpret_cast<SauvolaJob *>(b->hctrl + L.jobs);
        SauvolaJob *dj = reinterpret_cast<SauvolaJob *>(dctrl + L.jobs);
        unsigned *dcounts = reinterpret_cast<unsigned *>(dctrl + L.counts);
        for (int i = 0; i < nb; i++) {
            const BoxInfo &bi = b->boxes[i];
            hj[i].src = b->gray.page(bi.page) + (size_t)bi.t * b->gray.pitch + bi.l;
            hj[i].src_pitch = b->gray.pitch;
            hj[i].w = bi.r - bi.l; hj[i].h = bi.b - bi.t;
            hj[i].dst = b->thA.as<uint8_t>() + 256 + bi.off + bi.phase;
            hj[i].dst_inv = b->thB.as<uint8_t>() + 256 + bi.off + bi.phase;
            hj[i].dst_pitch = bi.pitch;
            hj[i].counts = dcounts + 2 * i;
            hj[i].bits = nullptr; hj[i].bits_pitch = 0; hj[i].no_bytes = 0;
        }
        HIP_TRY(hipMemsetAsync(dcounts, 0, (size_t)nb * 8, s));
        TRY(upload_1d(s, dj, hj, (size_t)nb * sizeof(SauvolaJob)));
        TRY(launch_sauvola_dev(ctx, s, hj, dj, nb, window, window, 0.1, 128.0, SAUVOLA_INVERT));          // mrc.py:229-235
        TRY(download_1d(s, b->hctrl + L.counts, dcounts, (size_t)nb * 8));
    }
    // ---- noise estimate of the central crop (mrc.py:280-292) ----
    SigJob *hsj = reinterpret_cast<SigJob *>(b->hctrl + L.sigjobs);
    SigJob *dsj = reinterpret_cast<SigJob *>(dctrl + L.sigjobs);
    for (int i = 0; i < N; i++) {
        hsj[i].src = b->gray.page(i) + (size_t)b->crop[0] * b->gray.pitch + b->crop[2];
        hsj[i].pitch = b->gray.pitch;
        hsj[i].w = b->crop[3] - b->crop[2]; hsj[i].h = b->crop[1] - b->crop[0];
        hsj[i].as_bool = 0;
        hsj[i].scratch = b->sig_scratch.as<char>() + (size_t)i * b->sig_stride;
    }
    TRY(upload_1d(s, dsj, hsj, (size_t)N * sizeof(SigJob)));
    double *dsig = reinterpret_cast<double *>(dctrl + L.sigma);
    TRY(launch_estimate_sigma_jobs(ctx, s, hsj, dsj, N, 0, dsig));
    TRY(download_1d(s, b->hctrl + L.sigma, dsig, (size_t)N * 8));
    b->state = 2;
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_sigmas(mrchip_batch *b, double *sigma_est) {
    CHECK_B(b);
    if (b->state < 2) { set_error("sigmas before mask_begin"); return MRCHIP_E_STATE; }
    HIP_TRY(hipStreamSynchronize(b->s));
    const CtrlLayout L(b->n, b->nb_cap);
    const double *hs = reinterpret_cast<const double *>(b->hctrl + L.sigma);
    for (int i = 0; i < b->n; i++) {
        b->sigma[i] = hs[i];
        if (sigma_est) sigma_est[i] = hs[i];
    }
    if (b->state == 2) b->state = 3;
    return 0;
}

// decisions (mrc.py:240-263, host float64 like the reference) + ordered commit (mrc.py:266).
// Boxes whose ratios do not settle the polarity need mean_estimate_sigma of both bool thresholds
// (mrc.py:253-254): all of them, over the whole batch, run as ONE job list (float64 path).
// Phase 1 (box_decisions_begin) settles what the ratios settle and enqueues those jobs + the copy of
// their results, marked by an event; phase 2 (box_decisions_commit) waits for the event only -- the
// caller puts independent work (blur, page threshold) on the stream in between -- and commits.
static int box_decisions_begin(mrchip_batch *b) {
    const int nb = (int)b->boxes.size();
    b->need.clear();
    if (nb == 0) return 0;
    mrchip_ctx *ctx = b->ctx;
    hipStream_t s = b->s;
    const CtrlLayout L(b->n, b->nb_cap);
    unsigned char *dctrl = b->ctrl.as<unsigned char>();
    const unsigned *counts = reinterpret_cast<const unsigned *>(b->hctrl + L.counts);
    std::vector<int> &need = b->need;          // boxes on the sigma path
    std::vector<double> &ratio = b->ratio, &inv_ratio = b->inv_ratio;
    ratio.assign(nb, 0.0); inv_ratio.assign(nb, 0.0);
    for (int i = 0; i < nb; i++) {
        BoxInfo &bi = b->boxes[i];
        const double size = (double)(bi.r - bi.l) * (double)(bi.b - bi.t);
        ratio[i] = (double)counts[2 * i] / size;                       // mrc.py:231-233
        inv_ratio[i] = (double)counts[2 * i + 1] / size;               // mrc.py:236-238
        bi.decision = 0;
        if (ratio[i] < 0.3 || inv_ratio[i] < 0.3) {                    // mrc.py:240
            if (inv_ratio[i] > 0.2 && ratio[i] < 0.2) bi.decision = 1; // mrc.py:247-248
            else need.push_back(i);
        }
    }
    if (!need.empty()) {
        const int nj = (int)need.size() * 2;
        size_t total = 0;
        std::vector<size_t> offs(nj);
        for (int k = 0; k < nj; k++) {
            const BoxInfo &bi = b->boxes[need[k / 2]];
            offs[k] = total;
            total += sigma_scratch_bytes(bi.r - bi.l, bi.b - bi.t, 1);
        }
        if (total + 256 > b->box_sig_cap) {          // grow-only: no synchronisation in the steady state
            HIP_TRY(hipStreamSynchronize(s));
            TRY(b->box_sig_scratch.alloc(ctx, total + total / 4 + 256));
            b->box_sig_cap = total + total / 4 + 256;
        }
        SigJob *hj = reinterpret_cast<SigJob *>(b->hctrl + L.boxsigjobs);
        SigJob *dj = reinterpret_cast<SigJob *>(dctrl + L.boxsigjobs);
        for (int k = 0; k < nj; k++) {
            const BoxInfo &bi = b->boxes[need[k / 2]];
            hj[k].src = ((k & 1) ? b->thB.as<uint8_t>() : b->thA.as<uint8_t>()) + 256 + bi.off + bi.phase;
            hj[k].pitch = bi.pitch;
            hj[k].w = bi.r - bi.l; hj[k].h = bi.b - bi.t;
            hj[k].as_bool = 1;
            hj[k].scratch = b->box_sig_scratch.as<char>() + offs[k];
        }
        double *dsig = reinterpret_cast<double *>(dctrl + L.box_sigma);
        TRY(upload_1d(s, dj, hj, (size_t)nj * sizeof(SigJob)));
        TRY(launch_estimate_sigma_jobs(ctx, s, hj, dj, nj, 1, dsig));
        TRY(download_1d(s, b->hctrl + L.box_sigma, dsig, (size_t)nj * 8));
        if (!b->box_ev) HIP_TRY(hipEventCreateWithFlags(&b->box_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(b->box_ev, s));
    }
    return 0;
}

// or_mode: OR the decided thresholds into a mask that already holds the page threshold (batch
// pipeline) instead of assigning them (mrc.create_hocr_mask on a caller's mask, mrc.py:266)
static int box_decisions_commit(mrchip_batch *b, int or_mode) {
    const int nb = (int)b->boxes.size();
    if (nb == 0) return 0;
    mrchip_ctx *ctx = b->ctx;
    hipStream_t s = b->s;
    const CtrlLayout L(b->n, b->nb_cap);
    unsigned char *dctrl = b->ctrl.as<unsigned char>();
    const std::vector<int> &need = b->need;
    const std::vector<double> &ratio = b->ratio, &inv_ratio = b->inv_ratio;
    if (!need.empty()) {
        HIP_TRY(hipEventSynchronize(b->box_ev));
        const double *hs = reinterpret_cast<const double *>(b->hctrl + L.box_sigma);
        for (size_t q = 0; q < need.size(); q++) {
            const int i = need[q];
            const double rs = hs[2 * q], irs = hs[2 * q + 1];          // mrc.py:253-254
            if (inv_ratio[i] < 0.3 && inv_ratio[i] < ratio[i] && (irs < rs || (rs < 0.1 && irs < 0.1)))
                b->boxes[i].decision = 2;
            else if (ratio[i] < 0.2)
                b->boxes[i].decision = 1;                              // mrc.py:258-263
        }
    }
    HocrBox *hb = reinterpret_cast<HocrBox *>(b->hctrl + L.boxes);
    HocrBox *db = reinterpret_cast<HocrBox *>(dctrl + L.boxes);
    int maxw = 0, maxh = 0;
    double area = 0;
    for (int i = 0; i < nb; i++) {
        const BoxInfo &bi = b->boxes[i];
        hb[i].l = bi.l; hb[i].t = bi.t; hb[i].r = bi.r; hb[i].b = bi.b;
        hb[i].decision = bi.decision;
        hb[i].th = b->thA.as<uint8_t>() + 256 + bi.off + bi.phase;
        hb[i].thi = b->thB.as<uint8_t>() + 256 + bi.off + bi.phase;
        hb[i].pitch = bi.pitch;
        hb[i].mask = b->mask.pl.page(bi.page);
        hb[i].mpitch = b->mask.pl.pitch;
        hb[i].page_end = b->first_box[bi.page + 1];
        hb[i].overlapped = 0;
        hb[i].bits = (or_mode && b->commit_bits)
                         ? reinterpret_cast<uint8_t *>(b->dn_bits.as<unsigned>() + (size_t)bi.page * b->dn_stride + (size_t)cdiv(b->w, 32) * b->h)
                         : nullptr;
        hb[i].bits_pitch = cdiv(b->w, 32) * 4;
        hb[i].no_bytes = hb[i].bits != nullptr;      // commit_bits: the denoiser's unpack rewrites the mask bytes from the bit rows
        for (int j = i + 1; j < hb[i].page_end; j++) {
            const BoxInfo &bj = b->boxes[j];
            if (bj.decision && bj.l < bi.r && bi.l < bj.r && bj.t < bi.b && bi.t < bj.b) { hb[i].overlapped = 1; break; }
        }
        maxw = std::max(maxw, bi.r - bi.l); maxh = std::max(maxh, bi.b - bi.t);
        if (bi.decision) area += (double)(bi.r - bi.l) * (double)(bi.b - bi.t);
    }
    TRY(upload_1d(s, db, hb, (size_t)nb * sizeof(HocrBox)));
    return launch_hocr_commit(ctx, s, db, nb, maxw, maxh, area, or_mode);
}

static int decide_and_commit(mrchip_batch *b) {       // assignment semantics, one go (mrchip_hocr_mask)
    TRY(box_decisions_begin(b));
    return box_decisions_commit(b, 0);
}

MRCHIP_EXPORT int mrchip_batch_mask_finish(mrchip_batch *b, const double *weights, const int *radius, int denoise_fast) {
    CHECK_B(b);
    if (b->state < 2) { set_error("mask_finish before mask_begin"); return MRCHIP_E_STATE; }
    if (b->state == 2) TRY(mrchip_batch_sigmas(b, nullptr));
    mrchip_ctx *ctx = b->ctx;
    hipStream_t s = b->s;
    const int w = b->w, h = b->h, N = b->active;
    const CtrlLayout L(b->n, b->nb_cap);
    unsigned char *dctrl = b->ctrl.as<unsigned char>();
    TRY(box_decisions_begin(b));          // box sigma jobs go first; their results are awaited after the page threshold is queued
    // ---- create_threshold_mask (mrc.py:300-329) ----
    GaussW *hg = reinterpret_cast<GaussW *>(b->hctrl + L.gauss);
    bool any_blur = false;
    int max_radius = 0;
    for (int i = 0; i < N; i++) {
        memset(&hg[i], 0, sizeof(GaussW));
        hg[i].w[0] = 1.0;                                               // radius 0: identity
        if (b->sigma[i] > 1.0) {                                        // mrc.py:309
            const double sg = b->sigma[i] * 0.1;
            const int r = (int)(4.0 * sg + 0.5);                        // scipy: int(truncate*sd + 0.5)
            if (r > GMAXR) { set_error("mask_finish: blur radius %d > %d", r, GMAXR); return MRCHIP_E_UNSUPPORTED; }
            if (weights) {
                if (!radius || radius[i] != r) {
                    set_error("mask_finish: page %d radius does not match sigma_est %.17g", i, b->sigma[i]);
                    return MRCHIP_E_ARG;
                }
                memcpy(hg[i].w, weights + (size_t)i * MRCHIP_MAX_TAPS, (size_t)(2 * r + 1) * sizeof(double));
            } else {
                std::vector<double> wl;
                TRY(gaussian_weights_libm(sg, wl));
                memcpy(hg[i].w, wl.data(), wl.size() * sizeof(double));
            }
            hg[i].radius = r;
            if (r > 0) any_blur = true;
            max_radius = std::max(max_radius, r);
        }
    }
    Plane thr_src = b->gray;
    if (any_blur) {
        bool fast_ok = true;
        for (int i = 0; i < N; i++) fast_ok = fast_ok && gauss_weights_allow_fast(hg[i]);
        if (gauss_uses_fused(w, h, max_radius))
            for (int i = 0; i < N; i++) gauss_pad_weights(hg[i], max_radius);
        GaussW *dg = reinterpret_cast<GaussW *>(dctrl + L.gauss);
        TRY(upload_1d(s, dg, hg, (size_t)N * sizeof(GaussW)));
        TRY(launch_gaussian_batch(ctx, s, b->gray, b->blur.pl, w, h, dg, b->gtmp.as<float>(), b->gtmp_pitch,
                                  b->gtmp_stride, N, max_radius, fast_ok));                              // mrc.py:311, 325
        thr_src = b->blur.pl;
    }
    SauvolaJob *hj = reinterpret_cast<SauvolaJob *>(b->hctrl + L.pjobs);
    SauvolaJob *dj = reinterpret_cast<SauvolaJob *>(dctrl + L.pjobs);
    // The denoiser works on 1-bpp rows.  Where the page launch takes the 8- / 16-column Sauvola kernel, that kernel and the
    // box commit write those rows alongside the mask bytes (a byte per lane more; a 16-bit OR per lane), and the separate
    // pass that re-read the whole byte mask to pack it is not run.
    const int wpr = cdiv(w, 32);
    const bool fuse_bits = denoise_fast && w > 4 && h > 4 && wpr <= 512 && sauvola_writes_bits(w, h, b->window);
    b->commit_bits = fuse_bits;
    for (int i = 0; i < N; i++) {
        hj[i].src = thr_src.page(i); hj[i].src_pitch = thr_src.pitch;
        hj[i].w = w; hj[i].h = h;
        hj[i].dst = b->mask.pl.page(i); hj[i].dst_pitch = b->mask.pl.pitch;
        hj[i].dst_inv = nullptr; hj[i].counts = nullptr;
        // the denoiser's ORIGINAL rows: second half of the page's bit scratch
        hj[i].bits = fuse_bits ? reinterpret_cast<uint8_t *>(b->dn_bits.as<unsigned>() + (size_t)i * b->dn_stride + (size_t)wpr * h) : nullptr;
        hj[i].bits_pitch = wpr * 4;
        // with the bit rows written here and by the commit, nothing reads the mask BYTES before the denoiser's unpack
        // rewrites them: they are not stored (1 byte per pixel less to write, 2 less to read-modify-write per box pixel)
        hj[i].no_bytes = fuse_bits ? 1 : 0;
    }
    TRY(upload_1d(s, dj, hj, (size_t)N * sizeof(SauvolaJob)));
    TRY(launch_sauvola_dev(ctx, s, hj, dj, N, b->window, b->window, 0.34, 128.0, SAUVOLA_INVERT));   // :325-329 (stored, not OR-ed)
    TRY(box_decisions_commit(b, 1));      // mrc.py:240-266 on top: mask = page threshold | box thresholds
    b->commit_bits = false;
    if (denoise_fast)
        TRY(launch_denoise_batch(ctx, s, b->mask.pl, w, h, 4, 2, b->dn_bits.as<unsigned>(), b->dn_stride, N, fuse_bits, true));  // :388
    b->state = 4;
    b->packed_valid = 0;
    // the denoiser's bit rows are the final mask when its bit-sliced path ran (launch_denoise_batch)
    b->bits_valid = denoise_fast && w > 4 && h > 4 && cdiv(w, 32) <= 512;
    return 0;
}

// mrc.threshold_image (mrc.py:58-87) of every page of the batch in one launch: Sauvola with a square window on
// the gray plane (the luma of an RGB page), True = dark stored into the mask plane; afterwards the mask can be
// downloaded like a finished one.  The Sauvola-only workload of BASELINE.json configs[2].
MRCHIP_EXPORT int mrchip_batch_threshold(mrchip_batch *b, int window, double k) {
    CHECK_B(b);
    if (b->state < 1) { set_error("threshold before upload"); return MRCHIP_E_STATE; }
    if (window < 1) { set_error("threshold: bad window"); return MRCHIP_E_ARG; }
    mrchip_ctx *ctx = b->ctx;
    hipStream_t s = b->s;
    const int N = b->active;
    TRY(ensure_ctrl(b, 0));
    const CtrlLayout L(b->n, b->nb_cap);
    if (b->c == 3) {
        for (int i = 0; i < N; i++)
            if (b->gray_given[i]) { set_error("threshold: caller-supplied gray planes are a create_mrc_hocr_components feature"); return MRCHIP_E_ARG; }
        TRY(launch_luma601(ctx, s, b->img.pl, b->gray_own.pl, b->w, b->h, N));
    }
    SauvolaJob *hj = reinterpret_cast<SauvolaJob *>(b->hctrl + L.pjobs);
    SauvolaJob *dj = reinterpret_cast<SauvolaJob *>(b->ctrl.as<unsigned char>() + L.pjobs);
    HIP_TRY(hipStreamSynchronize(s));           // the job array's previous copy may still be queued (no sigmas() in this flow)
    for (int i = 0; i < N; i++) {
        hj[i].src = b->gray.page(i); hj[i].src_pitch = b->gray.pitch;
        hj[i].w = b->w; hj[i].h = b->h;
        hj[i].dst = b->mask.pl.page(i); hj[i].dst_pitch = b->mask.pl.pitch;
        hj[i].dst_inv = nullptr; hj[i].counts = nullptr;
        hj[i].bits = nullptr; hj[i].bits_pitch = 0; hj[i].no_bytes = 0;
    }
    TRY(upload_1d(s, dj, hj, (size_t)N * sizeof(SauvolaJob)));
    TRY(launch_sauvola_dev(ctx, s, hj, dj, N, window, window, k, 128.0, SAUVOLA_INVERT));
    b->state = 4;
    b->bits_valid = 0;
    b->packed_valid = 0;
    b->layer_done[0] = b->layer_done[1] = 0;
    return 0;
}

static int download_mask_impl(mrchip_batch *b, int page, uint8_t *mask, bool wait) {
    CHECK_B(b);
    if (b->state < 4) { set_error("download_mask before mask_finish"); return MRCHIP_E_STATE; }
    if (page < 0 || page >= b->n || !mask) { set_error("download_mask: bad arguments"); return MRCHIP_E_ARG; }
    TRY(download_2d(b->s, mask, b->w, b->mask.pl.page(page), b->mask.pl.pitch, b->w, b->h));
    if (wait) HIP_TRY(hipStreamSynchronize(b->s));
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_download_mask(mrchip_batch *b, int page, uint8_t *mask) {
    return download_mask_impl(b, page, mask, true);
}
// enqueue only (see mrchip_batch_download_layer_async)
MRCHIP_EXPORT int mrchip_batch_download_mask_async(mrchip_batch *b, int page, uint8_t *mask) {
    return download_mask_impl(b, page, mask, false);
}

// The mask as the encoder wants it (mrc.py:474-520 makes a PIL mode '1' image of it): 1 bit per
// pixel, MSB first, (w+7)/8 bytes per row -- an eighth of the bytes over PCIe.
static int download_mask_packed_impl(mrchip_batch *b, int page, uint8_t *out, bool wait) {
    CHECK_B(b);
    if (b->state < 4) { set_error("download_mask_packed before mask_finish"); return MRCHIP_E_STATE; }
    if (page < 0 || page >= b->n || !out) { set_error("download_mask_packed: bad arguments"); return MRCHIP_E_ARG; }
    const size_t per_page = ((size_t)((b->w + 7) / 8) * b->h + 255) & ~(size_t)255;
    if (!b->packed_valid) {
        if (!b->packed.p) {
            HIP_TRY(hipStreamSynchronize(b->s));
            TRY(b->packed.alloc(b->ctx, per_page * b->n + 256));
        }
        TRY(launch_pack_msb(b->ctx, b->s, b->mask.pl, b->w, b->h, b->packed.as<uint8_t>(), per_page, b->active));
        b->packed_valid = 1;
    }
    TRY(download_1d(b->s, out, b->packed.as<uint8_t>() + per_page * page, (size_t)((b->w + 7) / 8) * b->h));
    if (wait) HIP_TRY(hipStreamSynchronize(b->s));
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_download_mask_packed(mrchip_batch *b, int page, uint8_t *out) {
    return download_mask_packed_impl(b, page, out, true);
}
MRCHIP_EXPORT int mrchip_batch_download_mask_packed_async(mrchip_batch *b, int page, uint8_t *out) {
    return download_mask_packed_impl(b, page, out, false);
}

// 1 when everything queued on the batch's stream has finished, 0 while work is pending (never blocks)
MRCHIP_EXPORT int mrchip_batch_done(mrchip_batch *b) {
    CHECK_B(b);
    const hipError_t e = hipStreamQuery(b->s);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    set_error("hipStreamQuery: %s", hipGetErrorString(e));
    return MRCHIP_E_HIP;
}

static int prepare_thumb(mrchip_batch *b, int Lr, double downsample, int *too_small) {
    mrchip_ctx *ctx = b->ctx;
    const int w = b->w, h = b->h, c = b->c, N = b->n;
    b->layer_w[Lr] = w; b->layer_h[Lr] = h; b->layer_small[Lr] = 0;
    if (too_small) *too_small = 0;
    if (!(downsample > 0)) return 0;
    const int wd = (int)(w / downsample), hd = (int)(h / downsample);                      // mrc.py:423-424
    if (wd <= 0 || hd <= 0) { if (too_small) *too_small = 1; return 0; }                   // mrc.py:429-431
    ThumbPlan &p = b->plan[Lr];
    if (p.w != w || p.h != h || p.c != c || b->plan_req[Lr][0] != wd || b->plan_req[Lr][1] != hd) {
        TRY(ThumbPlan_build(p, w, h, c, wd, hd));
        b->plan_req[Lr][0] = wd; b->plan_req[Lr][1] = hd;
        if (p.changed) {
            HIP_TRY(hipStreamSynchronize(b->s));
            TRY(b->small[Lr].alloc(ctx, N, p.ow * c, p.oh, false));
            TRY(b->sc1[Lr].alloc(ctx, N, p.rw * c, p.rh, false));
            int s2w, s2h;
            ThumbPlan_scratch2_dims(p, &s2w, &s2h);
            TRY(b->sc2[Lr].alloc(ctx, N, s2w, s2h, false));
            TRY(b->tables[Lr].alloc(ctx, ThumbPlan_table_bytes(p)));
            TRY(upload_1d(b->s, b->tables[Lr].p, p.blob_.data(), p.blob_.size()));      // (staged: ctx.hip; stream-ordered before the kernels)
        }
    }
    if (p.changed) { b->layer_w[Lr] = p.ow; b->layer_h[Lr] = p.oh; b->layer_small[Lr] = 1; }
    return 0;
}

// do_fg / do_bg: which layers to produce in this call (both in ONE optimise launch)
static int run_layers(mrchip_batch *b, bool do_fg, bool do_bg, double fg_ds, double bg_ds, int *too_small_fg,
                      int *too_small_bg) {
    mrchip_ctx *ctx = b->ctx;
    hipStream_t s = b->s;
    const int w = b->w, h = b->h, c = b->c, N = b->active;
    const CtrlLayout L(b->n, b->nb_cap);
    for (int Lr = 0; Lr < 2; Lr++) {
        if (!(Lr == 0 ? do_fg : do_bg)) continue;
        if (!b->layer[Lr].pl.p) TRY(b->layer[Lr].alloc(ctx, b->n, w * c, h));      // capacity, not the pages in use
        TRY(prepare_thumb(b, Lr, Lr == 0 ? fg_ds : bg_ds, Lr == 0 ? too_small_fg : too_small_bg));
    }
    OptJob *hj = reinterpret_cast<OptJob *>(b->hctrl + L.optjobs);
    OptJob *dj = reinterpret_cast<OptJob *>(b->ctrl.as<unsigned char>() + L.optjobs);
    int nj = 0, nmax = 0;
    for (int Lr = 0; Lr < 2; Lr++) {
        if (!(Lr == 0 ? do_fg : do_bg)) continue;
        for (int i = 0; i < N; i++) {
            OptJob &j = hj[nj++];
            j.mask = b->mask.pl.page(i); j.mpitch = b->mask.pl.pitch;
            j.mbits = b->bits_valid ? b->dn_bits.as<unsigned>() + (size_t)i * b->dn_stride : nullptr;
            j.mwpr = cdiv(w, 32);
            j.rowflags = b->bits_valid ? reinterpret_cast<const uint8_t *>(b->dn_bits.as<unsigned>() + (size_t)i * b->dn_stride) +
                                             denoise_rowflags_offset(w, h) : nullptr;
            j.img = b->img.pl.page(i); j.ipitch = b->img.pl.pitch;
            j.out = b->layer[Lr].pl.page(i); j.opitch = b->layer[Lr].pl.pitch;
            j.w = w; j.h = h;
            j.n = Lr ? 10 : 3;                                           // mrc.py:413/415, 447/449
            j.invert = Lr ? 1 : 0;                                       // mask_inv, mrc.py:439
            // a layer that only feeds its thumbnail need not hold the rows between the bands of its walkers (they are image
            // rows): the one-kernel thumbnail reads those from the image (honoured only if the band walkers take the launch)
            j.skip_copy = (b->layer_small[Lr] && j.mbits &&
                           thumbnail_reads_source_fused(b->plan[Lr], b->layer[Lr].pl, b->small[Lr].pl)) ? 1 : 0;
            j.rowmap = nullptr;
            nmax = std::max(nmax, j.n);
        }
    }
    TRY(launch_optimise_jobs(ctx, s, hj, dj, nj, w, h, c, nmax, &b->opt_mail));      // (uploads the job records)
    int first = 0;
    for (int Lr = 0; Lr < 2; Lr++) {
        if (!(Lr == 0 ? do_fg : do_bg)) continue;
        if (b->layer_small[Lr]) {
            // the launcher kept skip_copy where the band walkers ran: those layers hold their bands only
            ThumbAlt ta = {b->img.pl, hj[first].rowmap};
            const bool partial = hj[first].skip_copy && hj[first].rowmap;
            TRY(launch_thumbnail_plan(ctx, s, b->plan[Lr], b->layer[Lr].pl, b->small[Lr].pl, b->tables[Lr].p,
                                      b->sc1[Lr].pl, b->sc2[Lr].pl, N, partial ? &ta : nullptr));
        }
        b->layer_done[Lr] = 1;
        first += N;
    }
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_layers(mrchip_batch *b, int which, double fg_downsample, double bg_downsample,
                                      int *fg_w, int *fg_h, int *bg_w, int *bg_h, int *too_small) {
    CHECK_B(b);
    if (b->state < 4) { set_error("layers before mask_finish"); return MRCHIP_E_STATE; }
    if (!(which & 3)) { set_error("layers: which must select fg (1) and/or bg (2)"); return MRCHIP_E_ARG; }
    int ts_fg = 0, ts_bg = 0;
    TRY(run_layers(b, (which & 1) != 0, (which & 2) != 0, fg_downsample, bg_downsample, &ts_fg, &ts_bg));
    if (fg_w) *fg_w = b->layer_w[0];
    if (fg_h) *fg_h = b->layer_h[0];
    if (bg_w) *bg_w = b->layer_w[1];
    if (bg_h) *bg_h = b->layer_h[1];
    if (too_small) *too_small = (ts_fg ? 1 : 0) | (ts_bg ? 2 : 0);
    return 0;
}

static int download_layer_impl(mrchip_batch *b, int page, int is_bg, uint8_t *out, bool wait);

MRCHIP_EXPORT int mrchip_batch_download_layer(mrchip_batch *b, int page, int is_bg, uint8_t *out) {
    return download_layer_impl(b, page, is_bg, out, true);
}

// Enqueue only: the copy runs on the batch's stream behind the kernels that produce the layer; with `out` in
// pinned memory (mrchip_host_alloc) it is a true asynchronous DMA, so the host can hand page i to its encoder
// while page i+1 is still being decomposed / copied.  mrchip_batch_sync before reading.
MRCHIP_EXPORT int mrchip_batch_download_layer_async(mrchip_batch *b, int page, int is_bg, uint8_t *out) {
    return download_layer_impl(b, page, is_bg, out, false);
}

static int download_layer_impl(mrchip_batch *b, int page, int is_bg, uint8_t *out, bool wait) {
    CHECK_B(b);
    const int Lr = is_bg ? 1 : 0;
    if (!b->layer_done[Lr]) { set_error("download_layer before layers"); return MRCHIP_E_STATE; }
    if (page < 0 || page >= b->n || !out) { set_error("download_layer: bad arguments"); return MRCHIP_E_ARG; }
    const int c = b->c;
    if (b->layer_small[Lr]) {
        const size_t nbytes = (size_t)b->layer_w[Lr] * b->layer_h[Lr] * c;
        TRY(download_1d(b->s, out, b->small[Lr].pl.page(page), nbytes));
    } else {
        TRY(download_2d(b->s, out, b->w * c, b->layer[Lr].pl.page(page), b->layer[Lr].pl.pitch, b->w * c, b->h));
    }
    if (wait) {
        HIP_TRY(hipStreamSynchronize(b->s));
        TRY(optmail_check(&b->opt_mail));          // a strip hand-off of optimise timed out: the layer is not valid
    }
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_sync(mrchip_batch *b) {
    CHECK_B(b);
    HIP_TRY(hipStreamSynchronize(b->s));
    return optmail_check(&b->opt_mail);
}

MRCHIP_EXPORT int mrchip_batch_box_decisions(mrchip_batch *b, int page, int32_t *decisions, int nb) {
    CHECK_B(b);
    if (b->state < 4) { set_error("box_decisions before mask_finish"); return MRCHIP_E_STATE; }
    if (page < 0 || page >= b->n) { set_error("box_decisions: bad page"); return MRCHIP_E_ARG; }
    const int f = b->first_box[page], e = b->first_box[page + 1];
    for (int i = 0; i < nb && f + i < e; i++) decisions[i] = b->boxes[f + i].decision;
    return 0;
}

MRCHIP_EXPORT int mrchip_batch_device_ptrs(mrchip_batch *b, int page, void **img, void **mask, size_t *mask_pitch,
                                           void **fg, void **bg) {
    CHECK_B(b);
    if (page < 0 || page >= b->n) { set_error("device_ptrs: bad page"); return MRCHIP_E_ARG; }
    if (img) *img = b->img.pl.page(page);
    if (mask) *mask = b->mask.pl.page(page);
    if (mask_pitch) *mask_pitch = (size_t)b->mask.pl.pitch;
    if (fg) *fg = b->layer_small[0] ? b->small[0].pl.page(page) : (b->layer[0].pl.p ? b->layer[0].pl.page(page) : nullptr);
    if (bg) *bg = b->layer_small[1] ? b->small[1].pl.page(page) : (b->layer[1].pl.p ? b->layer[1].pl.page(page) : nullptr);
    return 0;
}

// ---- single page = batch of one ----------------------------------------------------
struct mrchip_page { mrchip_batch b; };

MRCHIP_EXPORT mrchip_page *mrchip_page_create(mrchip_ctx *ctx, int w, int h, int channels) {
    return reinterpret_cast<mrchip_page *>(mrchip_batch_create(ctx, 1, w, h, channels));
}
MRCHIP_EXPORT void mrchip_page_destroy(mrchip_page *pg) { mrchip_batch_destroy(reinterpret_cast<mrchip_batch *>(pg)); }
MRCHIP_EXPORT int mrchip_page_upload(mrchip_page *pg, const uint8_t *img) {
    return mrchip_batch_upload(reinterpret_cast<mrchip_batch *>(pg), 0, img);
}
MRCHIP_EXPORT int mrchip_page_upload_gray(mrchip_page *pg, const uint8_t *gray) {
    return mrchip_batch_upload_gray(reinterpret_cast<mrchip_batch *>(pg), 0, gray);
}
MRCHIP_EXPORT int mrchip_page_upload_mask(mrchip_page *pg, const uint8_t *mask) {
    return mrchip_batch_upload_mask(reinterpret_cast<mrchip_batch *>(pg), 0, mask);
}
MRCHIP_EXPORT int mrchip_page_mask_begin(mrchip_page *pg, const int32_t *boxes, int nb, int window) {
    mrchip_batch *b = reinterpret_cast<mrchip_batch *>(pg);
    TRY(mrchip_batch_set_boxes(b, 0, boxes, nb));
    return mrchip_batch_mask_begin(b, window);
}
MRCHIP_EXPORT int mrchip_page_sigma(mrchip_page *pg, double *sigma_est) {
    return mrchip_batch_sigmas(reinterpret_cast<mrchip_batch *>(pg), sigma_est);
}
MRCHIP_EXPORT int mrchip_page_mask_finish(mrchip_page *pg, const double *weights, int radius, int denoise_fast) {
    mrchip_batch *b = reinterpret_cast<mrchip_batch *>(pg);
    if (!b) { set_error("null page"); return MRCHIP_E_ARG; }
    if (weights) {
        if (radius < 0 || radius > GMAXR) { set_error("mask_finish: bad radius %d", radius); return MRCHIP_E_ARG; }
        double tab[MRCHIP_MAX_TAPS] = {0};
        memcpy(tab, weights, (size_t)(2 * radius + 1) * sizeof(double));
        return mrchip_batch_mask_finish(b, tab, &radius, denoise_fast);
    }
    return mrchip_batch_mask_finish(b, nullptr, nullptr, denoise_fast);
}
MRCHIP_EXPORT int mrchip_page_download_mask(mrchip_page *pg, uint8_t *mask) {
    return mrchip_batch_download_mask(reinterpret_cast<mrchip_batch *>(pg), 0, mask);
}
MRCHIP_EXPORT int mrchip_page_download_mask_packed(mrchip_page *pg, uint8_t *out) {
    return mrchip_batch_download_mask_packed(reinterpret_cast<mrchip_batch *>(pg), 0, out);
}
MRCHIP_EXPORT int mrchip_page_layer(mrchip_page *pg, int is_bg, double downsample, int *out_w, int *out_h, int *too_small) {
    mrchip_batch *b = reinterpret_cast<mrchip_batch *>(pg);
    int ts = 0;
    int rc = is_bg ? mrchip_batch_layers(b, 2, 0, downsample, nullptr, nullptr, out_w, out_h, &ts)
                   : mrchip_batch_layers(b, 1, downsample, 0, out_w, out_h, nullptr, nullptr, &ts);
    if (too_small) *too_small = ts ? 1 : 0;
    return rc;
}
// both layers in ONE launch (fg and bg page-layers side by side on the chip): what the generator does at its second
// yield -- the third then only downloads -- because one page alone is latency-bound per launch
MRCHIP_EXPORT int mrchip_page_layers(mrchip_page *pg, double fg_downsample, double bg_downsample, int *fg_w, int *fg_h,
                                     int *bg_w, int *bg_h, int *too_small) {
    return mrchip_batch_layers(reinterpret_cast<mrchip_batch *>(pg), 3, fg_downsample, bg_downsample, fg_w, fg_h, bg_w, bg_h,
                               too_small);
}
MRCHIP_EXPORT int mrchip_page_download_layer(mrchip_page *pg, int is_bg, uint8_t *out) {
    return mrchip_batch_download_layer(reinterpret_cast<mrchip_batch *>(pg), 0, is_bg, out);
}
MRCHIP_EXPORT int mrchip_page_sync(mrchip_page *pg) { return mrchip_batch_sync(reinterpret_cast<mrchip_batch *>(pg)); }
MRCHIP_EXPORT int mrchip_page_box_decisions(mrchip_page *pg, int32_t *decisions, int nb) {
    return mrchip_batch_box_decisions(reinterpret_cast<mrchip_batch *>(pg), 0, decisions, nb);
}
MRCHIP_EXPORT int mrchip_page_device_ptrs(mrchip_page *pg, void **img, void **mask, size_t *mask_pitch, void **fg,
                                          void **bg) {
    return mrchip_batch_device_ptrs(reinterpret_cast<mrchip_batch *>(pg), 0, img, mask, mask_pitch, fg, bg);
}

// ---- host-buffer entry points built on the batch machinery --------------------------
MRCHIP_EXPORT int mrchip_hocr_mask(mrchip_ctx *ctx, const uint8_t *gray, uint8_t *mask, int w, int h,
                                   const int32_t *boxes, int nb, int window, int32_t *decisions) {
    if (!ctx || !gray || !mask) { set_error("hocr_mask: bad arguments"); return MRCHIP_E_ARG; }
    if (nb == 0) return 0;
    mrchip_batch *b = mrchip_batch_create(ctx, 1, w, h, 1);
    if (!b) return MRCHIP_E_NOMEM;
    int rc = mrchip_batch_upload(b, 0, gray);
    if (!rc) rc = mrchip_batch_set_boxes(b, 0, boxes, nb);
    if (!rc) rc = mrchip_batch_mask_begin(b, window);
    if (!rc) rc = mrchip_batch_sigmas(b, nullptr);
    // mask_arr is modified in place (mrc.py:266): start from the caller's mask
    if (!rc) rc = upload_2d(b->s, b->mask.pl.p, b->mask.pl.pitch, mask, w, w, h);
    if (!rc) rc = decide_and_commit(b);
    if (!rc) rc = download_2d(b->s, mask, w, b->mask.pl.p, b->mask.pl.pitch, w, h);
    if (!rc && hipStreamSynchronize(b->s) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = MRCHIP_E_HIP; }
    if (!rc && decisions)
        for (int i = 0; i < nb; i++) decisions[i] = b->boxes[i].decision;
    mrchip_batch_destroy(b);
    return rc;
}
