// Host-buffer entry points: the numpy-array surface of the reference's Cython
// modules and third-party stages (include/mrchip.h).  Each call stages its
// arguments into pitched device images on the context's first stream, runs the
// HIP kernels and copies the result back; it returns after the result is in the
// caller's buffer (the reference's functions are synchronous too).
#include "mrchip_internal.h"

using namespace mrchip;

#define CHECK_CTX(ctx)                                                      \
    do {                                                                    \
        if (!(ctx)) { set_error("null context"); return MRCHIP_E_ARG; }     \
        HIP_TRY(hipSetDevice((ctx)->device));                               \
    } while (0)

MRCHIP_EXPORT int mrchip_selftest_sauvola_quotients(mrchip_ctx *ctx, long long *mismatches) {
    CHECK_CTX(ctx);
    if (!mismatches) { set_error("selftest: bad arguments"); return MRCHIP_E_ARG; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf bad;
    TRY(bad.alloc(ctx, 256));
    TRY(sauvola_div_selftest(ctx, s, bad.as<unsigned long long>()));
    unsigned long long h = 0;
    TRY(download_1d(s, &h, bad.p, 8));
    HIP_TRY(hipStreamSynchronize(s));
    *mismatches = (long long)h;
    return 0;
}

MRCHIP_EXPORT int mrchip_selftest_sauvola_table(mrchip_ctx *ctx, double k, double R, long long *mismatches, long long *tested,
                                                int *table_bytes) {
    CHECK_CTX(ctx);
    if (!mismatches || !tested) { set_error("selftest: bad arguments"); return MRCHIP_E_ARG; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf bad;
    TRY(bad.alloc(ctx, 256));
    TRY(sauvola_table_selftest(ctx, s, k, R, bad.as<unsigned long long>(), table_bytes));
    unsigned long long h[2] = {0, 0};
    TRY(download_1d(s, h, bad.p, 16));
    HIP_TRY(hipStreamSynchronize(s));
    *mismatches = (long long)h[0];
    *tested = (long long)h[1];
    return 0;
}

MRCHIP_EXPORT int mrchip_selftest_gauss_fast(mrchip_ctx *ctx, const double *weights, int radius, long long *mismatches,
                                             double *max_error) {
    if (!ctx || !weights || !mismatches || radius < 1 || radius > 2) { set_error("selftest_gauss_fast: radius 1 or 2"); return MRCHIP_E_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf gw, res;
    TRY(gw.alloc(ctx, sizeof(GaussW)));
    TRY(res.alloc(ctx, 64));
    GaussW G;
    memset(&G, 0, sizeof(G));
    G.radius = radius;
    for (int i = 0; i < 2 * radius + 1; i++) G.w[i] = weights[i];
    TRY(upload_1d(s, gw.p, &G, sizeof(G)));
    TRY(gauss_fast_selftest(ctx, s, gw.as<GaussW>(), res.as<unsigned long long>(), res.as<unsigned>() + 4));
    unsigned long long h[4] = {0, 0, 0, 0};
    TRY(download_1d(s, h, res.p, 32));
    HIP_TRY(hipStreamSynchronize(s));
    *mismatches = (long long)h[0];
    if (max_error) { const unsigned bits = (unsigned)(h[2] & 0xffffffffu); float f; memcpy(&f, &bits, 4); *max_error = (double)f; }
    return 0;
}

MRCHIP_EXPORT int mrchip_selftest_optimise_quotients(mrchip_ctx *ctx, long long *mismatches) {
    CHECK_CTX(ctx);
    if (!mismatches) { set_error("selftest: bad arguments"); return MRCHIP_E_ARG; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf bad;
    TRY(bad.alloc(ctx, 256));
    TRY(optimise_div_selftest(ctx, s, bad.as<unsigned long long>()));
    unsigned long long h = 0;
    TRY(download_1d(s, &h, bad.p, 8));
    HIP_TRY(hipStreamSynchronize(s));
    *mismatches = (long long)h;
    return 0;
}

MRCHIP_EXPORT int mrchip_window_for_dpi(int has_dpi, double dpi) {
    int window = 51;                          // mrc.py:68
    if (has_dpi) {
        window = (int)(dpi / 4);              // mrc.py:71
        if (window % 2 == 0) window += 1;     // mrc.py:72-73
    }
    return window;
}

MRCHIP_EXPORT int mrchip_sauvola_u8(mrchip_ctx *ctx, const uint8_t *in, uint8_t *out, int w, int h,
                                    int window_w, int window_h, double k, double R, int invert) {
    CHECK_CTX(ctx);
    if (!in || !out || w < 0 || h < 0) { set_error("sauvola: bad arguments"); return MRCHIP_E_ARG; }
    if (w == 0 || h == 0) return 0;
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 src, dst;
    TRY(src.alloc(ctx, w, h));
    TRY(dst.alloc(ctx, w, h));
    TRY(upload_2d(s, src.p, src.pitch, in, w, w, h));
    SauvolaJob job = {src.p, src.pitch, w, h, dst.p, dst.pitch, nullptr, nullptr};
    TRY(launch_sauvola(ctx, s, &job, 1, window_w, window_h, k, R, invert ? SAUVOLA_INVERT : 0));
    TRY(download_2d(s, out, w, dst.p, dst.pitch, w, h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_luma601(mrchip_ctx *ctx, const uint8_t *rgb, uint8_t *gray, int w, int h) {
    CHECK_CTX(ctx);
    if (!rgb || !gray || w < 0 || h < 0) { set_error("luma601: bad arguments"); return MRCHIP_E_ARG; }
    if (w == 0 || h == 0) return 0;
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 src, dst;
    TRY(src.alloc(ctx, w, h, 3));
    TRY(dst.alloc(ctx, w, h));
    TRY(upload_2d(s, src.p, src.pitch, rgb, w * 3, w * 3, h));
    Plane ps, pd;
    ps.p = src.p; ps.pitch = src.pitch; pd.p = dst.p; pd.pitch = dst.pitch;
    TRY(launch_luma601(ctx, s, ps, pd, w, h, 1));
    TRY(download_2d(s, gray, w, dst.p, dst.pitch, w, h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_mask_denoise(mrchip_ctx *ctx, uint8_t *mask, int w, int h, int mincnt, int n_size) {
    CHECK_CTX(ctx);
    if (!mask || w < 0 || h < 0) { set_error("mask_denoise: bad arguments"); return MRCHIP_E_ARG; }
    if (w == 0 || h == 0) return 0;
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 m;
    DevBuf bits;
    TRY(m.alloc(ctx, w, h));
    TRY(bits.alloc(ctx, denoise_scratch_bytes(w, h)));
    TRY(upload_2d(s, m.p, m.pitch, mask, w, w, h));
    Plane pm;
    pm.p = m.p; pm.pitch = m.pitch;
    TRY(launch_denoise_batch(ctx, s, pm, w, h, mincnt, n_size, bits.as<unsigned>(), 0, 1));
    TRY(download_2d(s, mask, w, m.p, m.pitch, w, h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_optimise(mrchip_ctx *ctx, const uint8_t *mask, const uint8_t *img, uint8_t *out,
                                  int w, int h, int channels, int n_size, int invert_mask) {
    CHECK_CTX(ctx);
    if (!mask || !img || !out || w < 0 || h < 0) { set_error("optimise: bad arguments"); return MRCHIP_E_ARG; }
    if (channels != 1 && channels != 3) { set_error("optimise: channels must be 1 or 3"); return MRCHIP_E_ARG; }
    if (w == 0 || h == 0) return 0;
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 m, i, o;
    DevBuf jb;
    TRY(m.alloc(ctx, w, h));
    TRY(i.alloc(ctx, w, h, channels));
    TRY(o.alloc(ctx, w, h, channels));
    TRY(jb.alloc(ctx, sizeof(OptJob)));
    TRY(upload_2d(s, m.p, m.pitch, mask, w, w, h));
    TRY(upload_2d(s, i.p, i.pitch, img, w * channels, w * channels, h));
    OptJob job = {m.p, m.pitch, i.p, i.pitch, o.p, o.pitch, w, h, n_size, invert_mask ? 1 : 0};
    job.mbits = nullptr; job.mwpr = 0; job.rowflags = nullptr; job.skip_copy = 0; job.rowmap = nullptr;
    if (!ctx->host_mail) ctx->host_mail = new OptMail;       // lives with the context (hand-off buffers, queue, error word)
    OptMail &mail = *ctx->host_mail;
    TRY(launch_optimise_jobs(ctx, s, &job, jb.as<OptJob>(), 1, w, h, channels, n_size, &mail));
    TRY(download_2d(s, out, w * channels, o.p, o.pitch, w * channels, h));
    HIP_TRY(hipStreamSynchronize(s));      // `job` lives on this stack frame
    return optmail_check(&mail);
}

MRCHIP_EXPORT int mrchip_estimate_sigma(mrchip_ctx *ctx, const uint8_t *arr, int stride, int w, int h, int kind,
                                        double *sigma) {
    CHECK_CTX(ctx);
    if (!arr || !sigma || w <= 0 || h <= 0 || stride < w || (kind != 0 && kind != 1)) {
        set_error("estimate_sigma: bad arguments");
        return MRCHIP_E_ARG;
    }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 m;
    DevBuf scratch, res;
    TRY(m.alloc(ctx, w, h));
    TRY(scratch.alloc(ctx, sigma_scratch_bytes(w, h, kind)));
    TRY(res.alloc(ctx, 64));
    TRY(upload_2d(s, m.p, m.pitch, arr, stride, w, h));
    DevBuf jb;
    TRY(jb.alloc(ctx, sizeof(SigJob)));
    SigJob job = {m.p, m.pitch, w, h, kind, scratch.as<char>()};
    TRY(upload_1d(s, jb.p, &job, sizeof(job)));
    HIP_TRY(hipStreamSynchronize(s));      // `job` lives on this stack frame
    TRY(launch_estimate_sigma_jobs(ctx, s, &job, jb.as<SigJob>(), 1, kind, res.as<double>()));
    TRY(download_1d(s, sigma, res.p, sizeof(double)));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_estimate_noise_u8(mrchip_ctx *ctx, const uint8_t *gray, int w, int h, double *sigma) {
    if (!gray || w <= 0 || h <= 0) { set_error("estimate_noise: bad arguments"); return MRCHIP_E_ARG; }
    int hs = (int)(h / 2.0 - h / 4.0), he = (int)(h / 2.0 + h / 4.0);      // mrc.py:282-285
    int ws = (int)(w / 2.0 - w / 4.0), we = (int)(w / 2.0 + w / 4.0);
    if (he == 0 || we == 0) { hs = 0; he = h; ws = 0; we = w; }           // mrc.py:288-292
    return mrchip_estimate_sigma(ctx, gray + (size_t)hs * w + ws, w, we - ws, he - hs, 0, sigma);
}

// ---- the same two stages on a float32 image that does not hold whole numbers 0..255 (mrc.py:273-329 take any float32
// image; the production path, mrc.py:372, always passes float32(gray) and uses the uint8 entry points above) -------------
MRCHIP_EXPORT int mrchip_estimate_sigma_f32(mrchip_ctx *ctx, const float *arr, int stride, int w, int h, double *sigma) {
    CHECK_CTX(ctx);
    if (!arr || !sigma || w <= 0 || h <= 0 || stride < w) { set_error("estimate_sigma_f32: bad arguments"); return MRCHIP_E_ARG; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf plane, scratch, res, jb;
    const int pitch = round_up(w * 4 + 64, 64);                  // bytes
    TRY(plane.alloc(ctx, (size_t)pitch * h + 2 * PAD));
    TRY(scratch.alloc(ctx, sigma_scratch_bytes(w, h, 2)));
    TRY(res.alloc(ctx, 64));
    TRY(jb.alloc(ctx, sizeof(SigJob)));
    uint8_t *p0 = plane.as<uint8_t>() + PAD;
    TRY(upload_2d(s, p0, pitch, reinterpret_cast<const uint8_t *>(arr), stride * 4, w * 4, h));
    SigJob job = {p0, pitch, w, h, 0, scratch.as<char>()};
    TRY(upload_1d(s, jb.p, &job, sizeof(job)));
    HIP_TRY(hipStreamSynchronize(s));      // `job` lives on this stack frame
    TRY(launch_estimate_sigma_jobs(ctx, s, &job, jb.as<SigJob>(), 1, 2, res.as<double>()));
    TRY(download_1d(s, sigma, res.p, sizeof(double)));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_estimate_noise_f32(mrchip_ctx *ctx, const float *gray, int w, int h, double *sigma) {
    if (!gray || w <= 0 || h <= 0) { set_error("estimate_noise_f32: bad arguments"); return MRCHIP_E_ARG; }
    int hs = (int)(h / 2.0 - h / 4.0), he = (int)(h / 2.0 + h / 4.0);      // mrc.py:282-285
    int ws = (int)(w / 2.0 - w / 4.0), we = (int)(w / 2.0 + w / 4.0);
    if (he == 0 || we == 0) { hs = 0; he = h; ws = 0; we = w; }           // mrc.py:288-292
    return mrchip_estimate_sigma_f32(ctx, gray + (size_t)hs * w + ws, w, we - ws, he - hs, sigma);
}

MRCHIP_EXPORT int mrchip_gaussian_f32(mrchip_ctx *ctx, const float *in, uint8_t *out, int w, int h, double sigma,
                                      const double *weights, int radius) {
    CHECK_CTX(ctx);
    if (!in || !out || w <= 0 || h <= 0) { set_error("gaussian_f32: bad arguments"); return MRCHIP_E_ARG; }
    std::vector<double> wl;
    if (radius == 0 && !weights) {
        wl.assign(1, 1.0);                    // no blur: out = uint8(in), the astype of mrc.py:325 alone
        weights = wl.data();
    } else if (!weights) {
        if (!(sigma > 0)) { set_error("gaussian_f32: sigma must be positive"); return MRCHIP_E_ARG; }
        TRY(gaussian_weights_libm(sigma, wl));
        weights = wl.data();
        radius = (int)(wl.size() / 2);
    } else if (radius != (int)(4.0 * sigma + 0.5)) {
        set_error("gaussian_f32: radius %d does not match sigma %.17g (scipy: int(4*sigma+0.5))", radius, sigma);
        return MRCHIP_E_ARG;
    }
    if (radius < 0 || radius > GMAXR) { set_error("gaussian_f32: radius %d outside [0,%d]", radius, GMAXR); return MRCHIP_E_UNSUPPORTED; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    DevBuf src, tmp, gw;
    Img8 dst;
    const int fp = round_up(w + 16, 16);                         // floats per row
    TRY(src.alloc(ctx, (size_t)fp * h * sizeof(float)));
    TRY(tmp.alloc(ctx, (size_t)fp * h * sizeof(float)));
    TRY(dst.alloc(ctx, w, h));
    TRY(gw.alloc(ctx, sizeof(GaussW)));
    GaussW G;
    memset(&G, 0, sizeof(G));
    G.radius = radius;
    for (int i = 0; i < 2 * radius + 1; i++) G.w[i] = weights[i];
    TRY(upload_1d(s, gw.p, &G, sizeof(G)));
    TRY(upload_2d(s, src.as<uint8_t>(), fp * 4, reinterpret_cast<const uint8_t *>(in), w * 4, w * 4, h));
    HIP_TRY(hipStreamSynchronize(s));      // G lives on this stack frame
    Plane pd;
    pd.p = dst.p; pd.pitch = dst.pitch;
    TRY(launch_gaussian_f32(ctx, s, src.as<float>(), fp, pd, w, h, gw.as<GaussW>(), tmp.as<float>(), fp));
    TRY(download_2d(s, out, w, dst.p, dst.pitch, w, h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_gaussian_u8(mrchip_ctx *ctx, const uint8_t *gray, uint8_t *out, int w, int h, double sigma,
                                     const double *weights, int radius) {
    CHECK_CTX(ctx);
    if (!gray || !out || w <= 0 || h <= 0 || !(sigma > 0)) { set_error("gaussian: bad arguments"); return MRCHIP_E_ARG; }
    std::vector<double> wl;
    if (!weights) {
        TRY(gaussian_weights_libm(sigma, wl));
        weights = wl.data();
        radius = (int)(wl.size() / 2);
    } else if (radius != (int)(4.0 * sigma + 0.5)) {
        set_error("gaussian: radius %d does not match sigma %.17g (scipy: int(4*sigma+0.5))", radius, sigma);
        return MRCHIP_E_ARG;
    }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 a, b;
    DevBuf tmp;
    const int tp = round_up(w, 16);
    TRY(a.alloc(ctx, w, h));
    TRY(b.alloc(ctx, w, h));
    TRY(tmp.alloc(ctx, (size_t)tp * h * sizeof(float)));
    TRY(upload_2d(s, a.p, a.pitch, gray, w, w, h));
    if (radius < 0 || radius > GMAXR) { set_error("gaussian: radius %d outside [0,%d]", radius, GMAXR); return MRCHIP_E_UNSUPPORTED; }
    DevBuf gw;
    TRY(gw.alloc(ctx, sizeof(GaussW)));
    GaussW G;
    memset(&G, 0, sizeof(G));
    G.radius = radius;
    for (int i = 0; i < 2 * radius + 1; i++) G.w[i] = weights[i];
    const bool fast_ok = gauss_weights_allow_fast(G);
    if (gauss_uses_fused(w, h, radius)) gauss_pad_weights(G, radius);
    TRY(upload_1d(s, gw.p, &G, sizeof(G)));
    HIP_TRY(hipStreamSynchronize(s));
    Plane pa, pb;
    pa.p = a.p; pa.pitch = a.pitch; pb.p = b.p; pb.pitch = b.pitch;
    TRY(launch_gaussian_batch(ctx, s, pa, pb, w, h, gw.as<GaussW>(), tmp.as<float>(), tp, 0, 1, radius, fast_ok));
    TRY(download_2d(s, out, w, b.p, b.pitch, w, h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

MRCHIP_EXPORT int mrchip_thumbnail_size(int w, int h, int req_w, int req_h, int *out_w, int *out_h) {
    if (w <= 0 || h <= 0 || req_w <= 0 || req_h <= 0 || !out_w || !out_h) { set_error("thumbnail_size: bad arguments"); return MRCHIP_E_ARG; }
    return thumbnail_size(w, h, req_w, req_h, out_w, out_h);
}

MRCHIP_EXPORT int mrchip_thumbnail(mrchip_ctx *ctx, const uint8_t *in, int w, int h, int channels, int req_w, int req_h,
                                   uint8_t *out) {
    return mrchip_thumbnail_ex(ctx, in, w, h, channels, req_w, req_h, MRCHIP_FILTER_BICUBIC, 2.0, out);
}

MRCHIP_EXPORT int mrchip_thumbnail_ex(mrchip_ctx *ctx, const uint8_t *in, int w, int h, int channels, int req_w, int req_h,
                                      int filter, double reducing_gap, uint8_t *out) {
    CHECK_CTX(ctx);
    if (!in || !out || w <= 0 || h <= 0 || req_w <= 0 || req_h <= 0 || (channels != 1 && channels != 3)) {
        set_error("thumbnail: bad arguments");
        return MRCHIP_E_ARG;
    }
    ThumbPlan p;
    TRY(ThumbPlan_build(p, w, h, channels, req_w, req_h, filter, reducing_gap));
    if (!p.changed) { memcpy(out, in, (size_t)w * h * channels); return 0; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    Img8 src;
    DevBuf dst, s1, s2, tab;
    const int c = channels;
    TRY(src.alloc(ctx, w, h, c));
    TRY(dst.alloc(ctx, (size_t)p.ow * p.oh * c + 256));
    TRY(s1.alloc(ctx, (size_t)p.rw * p.rh * c + 256));
    int s2w, s2h;
    ThumbPlan_scratch2_dims(p, &s2w, &s2h);
    TRY(s2.alloc(ctx, (size_t)s2w * s2h + 256));
    TRY(tab.alloc(ctx, ThumbPlan_table_bytes(p)));
    TRY(upload_1d(s, tab.p, p.blob_.data(), p.blob_.size()));
    TRY(upload_2d(s, src.p, src.pitch, in, w * c, w * c, h));
    Plane psrc, pdst, p1, p2;
    psrc.p = src.p; psrc.pitch = src.pitch;
    pdst.p = dst.as<uint8_t>(); pdst.pitch = p.ow * c;
    p1.p = s1.as<uint8_t>(); p1.pitch = p.rw * c;
    p2.p = s2.as<uint8_t>(); p2.pitch = s2w;
    TRY(launch_thumbnail_plan(ctx, s, p, psrc, pdst, tab.p, p1, p2, 1));
    TRY(download_1d(s, out, dst.p, (size_t)p.ow * p.oh * c));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ---- internetarchivepdf/grayconvert.py:38-66 special_gray_convert (recode.py:362) ---------------------------------------
// Two calls because the reference's scalar arithmetic between the two passes (bright_adjust, the thresholds, the level
// tables) is host work in the host's numpy: `begin` uploads the page, leaves it on the device and returns the exact
// channel statistics; `finish` takes the tables the host derived from them and produces the gray page.
MRCHIP_EXPORT int mrchip_special_gray_begin(mrchip_ctx *ctx, const uint8_t *rgb, int w, int h, unsigned long long *stats) {
    CHECK_CTX(ctx);
    if (!rgb || !stats || w <= 0 || h <= 0) { set_error("special_gray_begin: bad arguments"); return MRCHIP_E_ARG; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    delete ctx->gray_pending;
    ctx->gray_pending = nullptr;
    GrayPending *g = new GrayPending;
    struct Guard { GrayPending *g; ~Guard() { delete g; } } guard{g};       // released on every early return
    TRY(g->src.alloc(ctx, w, h, 3));
    g->w = w; g->h = h;
    DevBuf st;
    TRY(st.alloc(ctx, 256));
    TRY(upload_2d(s, g->src.p, g->src.pitch, rgb, w * 3, w * 3, h));
    TRY(launch_rgb_stats(ctx, s, g->src.p, g->src.pitch, w, h, st.p));
    RgbStats hs;
    TRY(download_1d(s, &hs, st.p, sizeof(hs)));
    HIP_TRY(hipStreamSynchronize(s));
    for (int c = 0; c < 3; c++) {
        stats[c] = hs.mn[c]; stats[3 + c] = hs.mx[c]; stats[6 + c] = hs.sum[c]; stats[9 + c] = hs.sumsq[c];
    }
    ctx->gray_pending = g;
    guard.g = nullptr;
    return 0;
}

MRCHIP_EXPORT int mrchip_special_gray_finish(mrchip_ctx *ctx, const uint8_t *level_luts, const uint8_t *hsl_table, uint8_t *out) {
    CHECK_CTX(ctx);
    GrayPending *g = ctx->gray_pending;
    if (!g) { set_error("special_gray_finish: no page pending (mrchip_special_gray_begin first)"); return MRCHIP_E_STATE; }
    hipStream_t s = ctx->streams[0];
    ScratchSync scratch_guard(ctx, s);
    struct Guard { mrchip_ctx *c; ~Guard() { delete c->gray_pending; c->gray_pending = nullptr; } } guard{ctx};   // the page is consumed
    if (!level_luts || !hsl_table || !out) { set_error("special_gray_finish: bad arguments"); return MRCHIP_E_ARG; }
    Img8 dst;
    DevBuf tab;
    TRY(dst.alloc(ctx, g->w, g->h));
    TRY(tab.alloc(ctx, 3 * 256 + 256 * 256));
    TRY(upload_1d(s, tab.p, level_luts, 3 * 256));
    TRY(upload_1d(s, tab.as<uint8_t>() + 3 * 256, hsl_table, 256 * 256));
    TRY(launch_rgb_level_hsl(ctx, s, g->src.p, g->src.pitch, dst.p, dst.pitch, g->w, g->h, tab.as<uint8_t>()));
    TRY(download_2d(s, out, g->w, dst.p, dst.pitch, g->w, g->h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
