// Device-resident page: the three yields of mrc.create_mrc_hocr_components
// (reference: internetarchivepdf/mrc.py:334-471) with the page's pixels crossing PCIe once.
//
//   mask_begin  (enqueue)  luma (mrc.py:361); both hOCR-box thresholds + set-pixel counts
//                          (mrc.py:223-238); noise estimate of the central crop (mrc.py:305)
//   sigma       (sync)     sigma_est back to the host, which -- like scipy -- owns the
//                          Gaussian weight table
//   mask_finish (enqueue)  box decisions (mrc.py:240-263) + ordered commit (mrc.py:266);
//                          blur (mrc.py:309-311); Sauvola k=0.34 OR-ed into the mask
//                          (mrc.py:325-329); fast denoise (mrc.py:388)
//   layer       (enqueue)  optimise fg (n=3) / bg (n=10, inverted mask) (mrc.py:412-415,
//                          446-449) + thumbnail (mrc.py:420-434, 454-468)
#include <algorithm>
#include <cmath>

#include "mrchip_internal.h"

using namespace mrchip;

struct BoxInfo {
    int l, t, r, b;
    size_t off;       // byte offset of the box's scratch rows
    int pitch;
    int phase;
    int decision;
};

struct mrchip_page {
    mrchip_ctx *ctx = nullptr;
    hipStream_t s = nullptr;
    int w = 0, h = 0, c = 1;
    Img8 img, gray_own, blur, mask, fg, bg;
    Img8 *gray = nullptr;
    DevBuf gtmp;  int gtmp_pitch = 0;          // float32 scratch of the blur
    DevBuf sig_scratch, box_sig_scratch, dn_bits, ctrl;   // ctrl: jobs / boxes / counts / sigma
    DevBuf thA, thB;  size_t th_bytes = 0;
    DevBuf small[2], sc1[2], sc2[2], tables[2];
    ThumbPlan plan[2];
    int plan_req[2][2] = {{0, 0}, {0, 0}};
    int layer_w[2] = {0, 0}, layer_h[2] = {0, 0}, layer_small[2] = {0, 0}, layer_done[2] = {0, 0};
    // pinned host mirror of the control block
    unsigned char *hctrl = nullptr;
    size_t ctrl_bytes = 0;
    std::vector<BoxInfo> boxes;
    int window = 51;
    double sigma_est = 0;
    int state = 0;    // 0 created, 1 uploaded, 2 mask_begin, 3 sigma known, 4 mask done
};

// control block layout (device + pinned mirror)
static constexpr size_t CTRL_SIGMA = 0;                 // double sigma; double box_sigma[2]
static constexpr size_t CTRL_COUNTS = 64;               // unsigned counts[2*nb]
static size_t ctrl_jobs_off(int nb) { return (CTRL_COUNTS + (size_t)nb * 8 + 63) & ~(size_t)63; }
static size_t ctrl_boxes_off(int nb) { return (ctrl_jobs_off(nb) + (size_t)nb * sizeof(SauvolaJob) + 63) & ~(size_t)63; }
static size_t ctrl_total(int nb) { return ctrl_boxes_off(nb) + (size_t)nb * sizeof(HocrBox) + 64; }

#define CHECK_PG(pg)                                                        \
    do {                                                                    \
        if (!(pg)) { set_error("null page"); return MRCHIP_E_ARG; }         \
        HIP_TRY(hipSetDevice((pg)->ctx->device));                           \
    } while (0)

MRCHIP_EXPORT mrchip_page *mrchip_page_create(mrchip_ctx *ctx, int w, int h, int channels) {
    if (!ctx || w <= 0 || h <= 0 || (channels != 1 && channels != 3)) {
        set_error("page_create: bad arguments (w=%d h=%d channels=%d)", w, h, channels);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    mrchip_page *pg = new mrchip_page();
    pg->ctx = ctx;
    pg->s = ctx->streams[ctx->next_stream++ % NSTREAMS];
    pg->w = w; pg->h = h; pg->c = channels;
    int rc = pg->img.alloc(ctx, w, h, channels);
    if (!rc && channels == 3) rc = pg->gray_own.alloc(ctx, w, h);
    if (!rc) rc = pg->mask.alloc(ctx, w, h);
    if (!rc) rc = pg->blur.alloc(ctx, w, h);
    pg->gray = channels == 3 ? &pg->gray_own : &pg->img;
    pg->gtmp_pitch = round_up(w, 16);
    if (!rc) rc = pg->gtmp.alloc(ctx, (size_t)pg->gtmp_pitch * h * sizeof(float));
    int hs, he, ws, we;
    hs = (int)(h / 2.0 - h / 4.0); he = (int)(h / 2.0 + h / 4.0);
    ws = (int)(w / 2.0 - w / 4.0); we = (int)(w / 2.0 + w / 4.0);
    if (he == 0 || we == 0) { hs = 0; he = h; ws = 0; we = w; }
    if (!rc) rc = pg->sig_scratch.alloc(ctx, sigma_scratch_bytes(we - ws, he - hs, 0));
    if (!rc) rc = pg->dn_bits.alloc(ctx, denoise_scratch_bytes(w, h));
    if (rc) { delete pg; return nullptr; }
    return pg;
}

MRCHIP_EXPORT void mrchip_page_destroy(mrchip_page *pg) {
    if (!pg) return;
    (void)hipSetDevice(pg->ctx->device);
    (void)hipStreamSynchronize(pg->s);
    if (pg->hctrl) (void)hipHostFree(pg->hctrl);
    delete pg;
}

MRCHIP_EXPORT int mrchip_page_upload(mrchip_page *pg, const uint8_t *img) {
    CHECK_PG(pg);
    if (!img) { set_error("page_upload: null image"); return MRCHIP_E_ARG; }
    TRY(upload_2d(pg->s, pg->img.p, pg->img.pitch, img, pg->w * pg->c, pg->w * pg->c, pg->h));
    pg->state = 1;
    pg->layer_done[0] = pg->layer_done[1] = 0;
    return 0;
}

static int ensure_ctrl(mrchip_page *pg, int nb) {
    size_t need = ctrl_total(nb);
    if (need > pg->ctrl_bytes) {
        if (pg->hctrl) { HIP_TRY(hipStreamSynchronize(pg->s)); HIP_TRY(hipHostFree(pg->hctrl)); pg->hctrl = nullptr; }
        need = (need + 4095) & ~(size_t)4095;
        TRY(pg->ctrl.alloc(pg->ctx, need));
        HIP_TRY(hipHostMalloc((void **)&pg->hctrl, need, hipHostMallocDefault));
        pg->ctrl_bytes = need;
    }
    return 0;
}

MRCHIP_EXPORT int mrchip_page_mask_begin(mrchip_page *pg, const int32_t *boxes, int nb, int window) {
    CHECK_PG(pg);
    if (pg->state < 1) { set_error("page_mask_begin before page_upload"); return MRCHIP_E_STATE; }
    if (nb < 0 || (nb > 0 && !boxes) || window < 1) { set_error("page_mask_begin: bad arguments"); return MRCHIP_E_ARG; }
    mrchip_ctx *ctx = pg->ctx;
    hipStream_t s = pg->s;
    const int w = pg->w, h = pg->h;
    pg->window = window;
    if (pg->c == 3)
        TRY(launch_luma601(ctx, s, pg->img.p, pg->img.pitch, pg->gray_own.p, pg->gray_own.pitch, w, h));   // mrc.py:361
    HIP_TRY(hipMemsetAsync(pg->mask.p, 0, pg->mask.bytes(), s));                                          // mrc.py:367
    TRY(ensure_ctrl(pg, nb));
    // ---- hOCR boxes: both polarities + counts ----
    pg->boxes.clear();
    size_t off = 0;
    for (int i = 0; i < nb; i++) {
        BoxInfo b;
        b.l = boxes[4 * i]; b.t = boxes[4 * i + 1]; b.r = boxes[4 * i + 2]; b.b = boxes[4 * i + 3];
        if (b.l < 0 || b.t < 0 || b.r > w || b.b > h || b.l >= b.r || b.t >= b.b) {
            set_error("page_mask_begin: box %d (%d,%d,%d,%d) is not inside the %dx%d page (the caller filters, mrc.py:212-221)",
                      i, b.l, b.t, b.r, b.b, w, h);
            return MRCHIP_E_ARG;
        }
        b.phase = b.l & 15;
        b.pitch = round_up(b.r - b.l + b.phase, 16) + 16;
        b.off = off;
        b.decision = 0;
        off += (size_t)b.pitch * (b.b - b.t);
        pg->boxes.push_back(b);
    }
    if (nb > 0) {
        if (off + 4096 > pg->th_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            pg->th_bytes = off + 4096 + off / 4;
            TRY(pg->thA.alloc(ctx, pg->th_bytes));
            TRY(pg->thB.alloc(ctx, pg->th_bytes));
        }
        SauvolaJob *hj = reinterpret_cast<SauvolaJob *>(pg->hctrl + ctrl_jobs_off(nb));
        SauvolaJob *dj = reinterpret_cast<SauvolaJob *>(pg->ctrl.as<unsigned char>() + ctrl_jobs_off(nb));
        unsigned *dcounts = reinterpret_cast<unsigned *>(pg->ctrl.as<unsigned char>() + CTRL_COUNTS);
        for (int i = 0; i < nb; i++) {
            const BoxInfo &b = pg->boxes[i];
            hj[i].src = pg->gray->p + (size_t)b.t * pg->gray->pitch + b.l;
            hj[i].src_pitch = pg->gray->pitch;
            hj[i].w = b.r - b.l; hj[i].h = b.b - b.t;
            hj[i].dst = pg->thA.as<uint8_t>() + 256 + b.off + b.phase;
            hj[i].dst_inv = pg->thB.as<uint8_t>() + 256 + b.off + b.phase;
            hj[i].dst_pitch = b.pitch;
            hj[i].counts = dcounts + 2 * i;
        }
        HIP_TRY(hipMemsetAsync(dcounts, 0, (size_t)nb * 8, s));
        HIP_TRY(hipMemcpyAsync(dj, hj, (size_t)nb * sizeof(SauvolaJob), hipMemcpyHostToDevice, s));
        TRY(launch_sauvola_dev(ctx, s, hj, dj, nb, window, window, 0.1, 128.0, SAUVOLA_INVERT));       // mrc.py:229-235
        HIP_TRY(hipMemcpyAsync(pg->hctrl + CTRL_COUNTS, dcounts, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
    }
    // ---- noise estimate of the central crop (mrc.py:280-292) ----
    int hs = (int)(h / 2.0 - h / 4.0), he = (int)(h / 2.0 + h / 4.0);
    int ws = (int)(w / 2.0 - w / 4.0), we = (int)(w / 2.0 + w / 4.0);
    if (he == 0 || we == 0) { hs = 0; he = h; ws = 0; we = w; }
    double *dsig = reinterpret_cast<double *>(pg->ctrl.as<unsigned char>() + CTRL_SIGMA);
    TRY(launch_estimate_sigma_scratch(ctx, s, pg->gray->p + (size_t)hs * pg->gray->pitch + ws, pg->gray->pitch,
                                      we - ws, he - hs, 0, dsig, pg->sig_scratch.p));
    HIP_TRY(hipMemcpyAsync(pg->hctrl + CTRL_SIGMA, dsig, sizeof(double), hipMemcpyDeviceToHost, s));
    pg->state = 2;
    return 0;
}

MRCHIP_EXPORT int mrchip_page_sigma(mrchip_page *pg, double *sigma_est) {
    CHECK_PG(pg);
    if (pg->state < 2) { set_error("page_sigma before page_mask_begin"); return MRCHIP_E_STATE; }
    HIP_TRY(hipStreamSynchronize(pg->s));
    pg->sigma_est = *reinterpret_cast<double *>(pg->hctrl + CTRL_SIGMA);
    if (sigma_est) *sigma_est = pg->sigma_est;
    if (pg->state == 2) pg->state = 3;
    return 0;
}

// mean_estimate_sigma of one of a box's bool thresholds (mrc.py:253-254); synchronous, rare
static int box_sigma(mrchip_page *pg, const BoxInfo &b, int inv, double *out) {
    mrchip_ctx *ctx = pg->ctx;
    const int bw = b.r - b.l, bh = b.b - b.t;
    const size_t need = sigma_scratch_bytes(bw, bh, 1);
    HIP_TRY(hipStreamSynchronize(pg->s));
    TRY(pg->box_sig_scratch.alloc(ctx, need));
    const uint8_t *src = (inv ? pg->thB.as<uint8_t>() : pg->thA.as<uint8_t>()) + 256 + b.off + b.phase;
    double *dsig = reinterpret_cast<double *>(pg->ctrl.as<unsigned char>() + CTRL_SIGMA) + 1;
    TRY(launch_estimate_sigma_scratch(ctx, pg->s, src, b.pitch, bw, bh, 1, dsig, pg->box_sig_scratch.p));
    HIP_TRY(hipMemcpyAsync(pg->hctrl + CTRL_SIGMA + 8, dsig, sizeof(double), hipMemcpyDeviceToHost, pg->s));
    HIP_TRY(hipStreamSynchronize(pg->s));
    *out = *reinterpret_cast<double *>(pg->hctrl + CTRL_SIGMA + 8);
    return 0;
}

MRCHIP_EXPORT int mrchip_page_mask_finish(mrchip_page *pg, const double *weights, int radius, int denoise_fast) {
    CHECK_PG(pg);
    if (pg->state < 2) { set_error("page_mask_finish before page_mask_begin"); return MRCHIP_E_STATE; }
    if (pg->state == 2) TRY(mrchip_page_sigma(pg, nullptr));
    mrchip_ctx *ctx = pg->ctx;
    hipStream_t s = pg->s;
    const int w = pg->w, h = pg->h, nb = (int)pg->boxes.size();
    // ---- box decisions: the ratio tree of mrc.py:240-263 (host, float64 like the reference) ----
    if (nb > 0) {
        const unsigned *counts = reinterpret_cast<const unsigned *>(pg->hctrl + CTRL_COUNTS);
        HocrBox *hb = reinterpret_cast<HocrBox *>(pg->hctrl + ctrl_boxes_off(nb));
        HocrBox *db = reinterpret_cast<HocrBox *>(pg->ctrl.as<unsigned char>() + ctrl_boxes_off(nb));
        int maxw = 0, maxh = 0;
        double area = 0;
        for (int i = 0; i < nb; i++) {
            BoxInfo &b = pg->boxes[i];
            const double size = (double)(b.r - b.l) * (double)(b.b - b.t);
            const double ratio = (double)counts[2 * i] / size;             // mrc.py:231-233
            const double inv_ratio = (double)counts[2 * i + 1] / size;     // mrc.py:236-238
            int dec = 0;
            if (ratio < 0.3 || inv_ratio < 0.3) {                          // mrc.py:240
                if (inv_ratio > 0.2 && ratio < 0.2) dec = 1;               // mrc.py:247-248
                else {
                    double rs = 0, irs = 0;
                    TRY(box_sigma(pg, b, 0, &rs));                         // mrc.py:253
                    TRY(box_sigma(pg, b, 1, &irs));                        // mrc.py:254
                    if (inv_ratio < 0.3 && inv_ratio < ratio && (irs < rs || (rs < 0.1 && irs < 0.1))) dec = 2;
                    else if (ratio < 0.2) dec = 1;                         // mrc.py:258-263
                }
            }
            b.decision = dec;
            hb[i].l = b.l; hb[i].t = b.t; hb[i].r = b.r; hb[i].b = b.b;
            hb[i].decision = dec;
            hb[i].th = pg->thA.as<uint8_t>() + 256 + b.off + b.phase;
            hb[i].thi = pg->thB.as<uint8_t>() + 256 + b.off + b.phase;
            hb[i].pitch = b.pitch;
            maxw = std::max(maxw, b.r - b.l); maxh = std::max(maxh, b.b - b.t);
            if (dec) area += size;
        }
        HIP_TRY(hipMemcpyAsync(db, hb, (size_t)nb * sizeof(HocrBox), hipMemcpyHostToDevice, s));
        TRY(launch_hocr_commit(ctx, s, pg->mask.p, pg->mask.pitch, db, nb, maxw, maxh, area));           // mrc.py:266
    }
    // ---- create_threshold_mask (mrc.py:300-329) ----
    const Img8 *thr_src = pg->gray;
    if (pg->sigma_est > 1.0) {                                                                            // mrc.py:309
        std::vector<double> wl;
        if (!weights) {
            TRY(gaussian_weights_libm(pg->sigma_est * 0.1, wl));
            weights = wl.data();
            radius = (int)(wl.size() / 2);
        } else if (radius != (int)(4.0 * (pg->sigma_est * 0.1) + 0.5)) {
            set_error("page_mask_finish: radius %d does not match sigma_est %.17g", radius, pg->sigma_est);
            return MRCHIP_E_ARG;
        }
        TRY(launch_gaussian_u8_scratch(ctx, s, pg->gray->p, pg->gray->pitch, pg->blur.p, pg->blur.pitch, w, h, weights,
                                       radius, pg->gtmp.as<float>(), pg->gtmp_pitch));                   // mrc.py:311, 325
        thr_src = &pg->blur;
    }
    SauvolaJob job = {thr_src->p, thr_src->pitch, w, h, pg->mask.p, pg->mask.pitch, nullptr, nullptr};
    TRY(launch_sauvola(ctx, s, &job, 1, pg->window, pg->window, 0.34, 128.0, SAUVOLA_INVERT | SAUVOLA_OR)); // :325-329
    if (denoise_fast)
        TRY(launch_denoise_scratch(ctx, s, pg->mask.p, pg->mask.pitch, w, h, 4, 2, pg->dn_bits.as<unsigned>())); // :388
    pg->state = 4;
    return 0;
}

MRCHIP_EXPORT int mrchip_page_download_mask(mrchip_page *pg, uint8_t *mask) {
    CHECK_PG(pg);
    if (pg->state < 4) { set_error("page_download_mask before page_mask_finish"); return MRCHIP_E_STATE; }
    TRY(download_2d(pg->s, mask, pg->w, pg->mask.p, pg->mask.pitch, pg->w, pg->h));
    HIP_TRY(hipStreamSynchronize(pg->s));
    return 0;
}

MRCHIP_EXPORT int mrchip_page_layer(mrchip_page *pg, int is_bg, double downsample, int *out_w, int *out_h,
                                    int *too_small) {
    CHECK_PG(pg);
    if (pg->state < 4) { set_error("page_layer before page_mask_finish"); return MRCHIP_E_STATE; }
    mrchip_ctx *ctx = pg->ctx;
    hipStream_t s = pg->s;
    const int w = pg->w, h = pg->h, c = pg->c, L = is_bg ? 1 : 0;
    Img8 &full = is_bg ? pg->bg : pg->fg;
    if (!full.p) TRY(full.alloc(ctx, w, h, c));
    TRY(launch_optimise(ctx, s, pg->mask.p, pg->mask.pitch, pg->img.p, pg->img.pitch, full.p, full.pitch, w, h, c,
                        is_bg ? 10 : 3, is_bg ? 1 : 0));                                      // mrc.py:412-415, 446-449
    pg->layer_w[L] = w; pg->layer_h[L] = h; pg->layer_small[L] = 0;
    if (too_small) *too_small = 0;
    if (downsample > 0) {
        const int wd = (int)(w / downsample), hd = (int)(h / downsample);                      // mrc.py:423-424
        if (wd > 0 && hd > 0) {
            ThumbPlan &p = pg->plan[L];
            if (p.w != w || p.h != h || p.c != c || pg->plan_req[L][0] != wd || pg->plan_req[L][1] != hd) {
                TRY(ThumbPlan_build(p, w, h, c, wd, hd));
                pg->plan_req[L][0] = wd; pg->plan_req[L][1] = hd;
                if (p.changed) {
                    HIP_TRY(hipStreamSynchronize(s));
                    TRY(pg->small[L].alloc(ctx, (size_t)p.ow * p.oh * c + 256));
                    TRY(pg->sc1[L].alloc(ctx, (size_t)p.rw * p.rh * c + 256));
                    TRY(pg->sc2[L].alloc(ctx, (size_t)p.ow * p.rh * c + 256));
                    TRY(pg->tables[L].alloc(ctx, ThumbPlan_table_bytes(p)));
                    int32_t *d = pg->tables[L].as<int32_t>();
                    HIP_TRY(hipMemcpy(d, p.bh_.data(), p.bh_.size() * 4, hipMemcpyHostToDevice)); d += p.bh_.size();
                    HIP_TRY(hipMemcpy(d, p.kh_.data(), p.kh_.size() * 4, hipMemcpyHostToDevice)); d += p.kh_.size();
                    HIP_TRY(hipMemcpy(d, p.bv_.data(), p.bv_.size() * 4, hipMemcpyHostToDevice)); d += p.bv_.size();
                    HIP_TRY(hipMemcpy(d, p.kv_.data(), p.kv_.size() * 4, hipMemcpyHostToDevice));
                }
            }
            if (p.changed) {
                TRY(launch_thumbnail_plan(ctx, s, p, full.p, full.pitch, pg->small[L].as<uint8_t>(), p.ow * c,
                                          pg->tables[L].as<int32_t>(), pg->sc1[L].as<uint8_t>(), pg->sc2[L].as<uint8_t>()));
                pg->layer_w[L] = p.ow; pg->layer_h[L] = p.oh; pg->layer_small[L] = 1;
            } else {
                pg->layer_w[L] = w; pg->layer_h[L] = h;
            }
        } else if (too_small) {
            *too_small = 1;                                                                    // mrc.py:429-431
        }
    }
    if (out_w) *out_w = pg->layer_w[L];
    if (out_h) *out_h = pg->layer_h[L];
    pg->layer_done[L] = 1;
    return 0;
}

MRCHIP_EXPORT int mrchip_page_download_layer(mrchip_page *pg, int is_bg, uint8_t *out) {
    CHECK_PG(pg);
    const int L = is_bg ? 1 : 0;
    if (!pg->layer_done[L]) { set_error("page_download_layer before page_layer"); return MRCHIP_E_STATE; }
    const int c = pg->c;
    if (pg->layer_small[L]) {
        const size_t n = (size_t)pg->layer_w[L] * pg->layer_h[L] * c;
        HIP_TRY(hipMemcpyAsync(out, pg->small[L].p, n, hipMemcpyDeviceToHost, pg->s));
    } else {
        Img8 &full = is_bg ? pg->bg : pg->fg;
        TRY(download_2d(pg->s, out, pg->w * c, full.p, full.pitch, pg->w * c, pg->h));
    }
    HIP_TRY(hipStreamSynchronize(pg->s));
    return 0;
}

MRCHIP_EXPORT int mrchip_page_sync(mrchip_page *pg) {
    CHECK_PG(pg);
    HIP_TRY(hipStreamSynchronize(pg->s));
    return 0;
}

MRCHIP_EXPORT int mrchip_page_box_decisions(mrchip_page *pg, int32_t *decisions, int nb) {
    CHECK_PG(pg);
    if (pg->state < 4) { set_error("page_box_decisions before page_mask_finish"); return MRCHIP_E_STATE; }
    for (int i = 0; i < nb && i < (int)pg->boxes.size(); i++) decisions[i] = pg->boxes[i].decision;
    return 0;
}

MRCHIP_EXPORT int mrchip_page_device_ptrs(mrchip_page *pg, void **img, void **mask, size_t *mask_pitch, void **fg,
                                          void **bg) {
    CHECK_PG(pg);
    if (img) *img = pg->img.p;
    if (mask) *mask = pg->mask.p;
    if (mask_pitch) *mask_pitch = (size_t)pg->mask.pitch;
    if (fg) *fg = pg->layer_small[0] ? pg->small[0].p : (void *)pg->fg.p;
    if (bg) *bg = pg->layer_small[1] ? pg->small[1].p : (void *)pg->bg.p;
    return 0;
}

// ---- host-buffer entry points built on the page machinery ------------------------
MRCHIP_EXPORT int mrchip_hocr_mask(mrchip_ctx *ctx, const uint8_t *gray, uint8_t *mask, int w, int h,
                                   const int32_t *boxes, int nb, int window, int32_t *decisions) {
    if (!ctx || !gray || !mask) { set_error("hocr_mask: bad arguments"); return MRCHIP_E_ARG; }
    if (nb == 0) return 0;
    mrchip_page *pg = mrchip_page_create(ctx, w, h, 1);
    if (!pg) return MRCHIP_E_NOMEM;
    int rc = mrchip_page_upload(pg, gray);
    hipStream_t s = pg->s;
    if (!rc) rc = mrchip_page_mask_begin(pg, boxes, nb, window);
    if (!rc) rc = mrchip_page_sigma(pg, nullptr);
    if (!rc) {
        // only the box part of mask_finish: start from the caller's mask (mask_arr is modified in place)
        rc = upload_2d(s, pg->mask.p, pg->mask.pitch, mask, w, w, h);
    }
    if (!rc) {
        // run decisions + commit by temporarily disabling the global threshold: done by hand here
        const unsigned *counts = reinterpret_cast<const unsigned *>(pg->hctrl + CTRL_COUNTS);
        HocrBox *hb = reinterpret_cast<HocrBox *>(pg->hctrl + ctrl_boxes_off(nb));
        HocrBox *db = reinterpret_cast<HocrBox *>(pg->ctrl.as<unsigned char>() + ctrl_boxes_off(nb));
        int maxw = 0, maxh = 0;
        double area = 0;
        for (int i = 0; i < nb && !rc; i++) {
            BoxInfo &b = pg->boxes[i];
            const double size = (double)(b.r - b.l) * (double)(b.b - b.t);
            const double ratio = (double)counts[2 * i] / size, inv_ratio = (double)counts[2 * i + 1] / size;
            int dec = 0;
            if (ratio < 0.3 || inv_ratio < 0.3) {
                if (inv_ratio > 0.2 && ratio < 0.2) dec = 1;
                else {
                    double rs = 0, irs = 0;
                    rc = box_sigma(pg, b, 0, &rs);
                    if (!rc) rc = box_sigma(pg, b, 1, &irs);
                    if (inv_ratio < 0.3 && inv_ratio < ratio && (irs < rs || (rs < 0.1 && irs < 0.1))) dec = 2;
                    else if (ratio < 0.2) dec = 1;
                }
            }
            b.decision = dec;
            if (decisions) decisions[i] = dec;
            hb[i].l = b.l; hb[i].t = b.t; hb[i].r = b.r; hb[i].b = b.b; hb[i].decision = dec;
            hb[i].th = pg->thA.as<uint8_t>() + 256 + b.off + b.phase;
            hb[i].thi = pg->thB.as<uint8_t>() + 256 + b.off + b.phase;
            hb[i].pitch = b.pitch;
            maxw = std::max(maxw, b.r - b.l); maxh = std::max(maxh, b.b - b.t);
            if (dec) area += size;
        }
        if (!rc && hipMemcpyAsync(db, hb, (size_t)nb * sizeof(HocrBox), hipMemcpyHostToDevice, s) != hipSuccess) rc = MRCHIP_E_HIP;
        if (!rc) rc = launch_hocr_commit(ctx, s, pg->mask.p, pg->mask.pitch, db, nb, maxw, maxh, area);
        if (!rc) rc = download_2d(s, mask, w, pg->mask.p, pg->mask.pitch, w, h);
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = MRCHIP_E_HIP;
    }
    mrchip_page_destroy(pg);
    return rc;
}
