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
    Img8 src, dst;
    TRY(src.alloc(ctx, w, h, 3));
    TRY(dst.alloc(ctx, w, h));
    TRY(upload_2d(s, src.p, src.pitch, rgb, w * 3, w * 3, h));
    TRY(launch_luma601(ctx, s, src.p, src.pitch, dst.p, dst.pitch, w, h));
    TRY(download_2d(s, gray, w, dst.p, dst.pitch, w, h));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
