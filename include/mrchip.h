/*
 * mrchip.h -- C ABI of libmrchip.so: the MI355X (gfx950) implementation of the
 * archive-pdf-tools MRC page-decomposition hot path.
 *
 * Every entry point replaces one interface of the reference (cited per
 * function as file:line under /root/reference).  Plain pointers and sizes only.
 * All functions return 0 on success or a negative MRCHIP_E_* code;
 * mrchip_last_error() returns a thread-local message.  There is NO CPU
 * fallback behind any of these: without a usable HIP device every call fails.
 *
 * Buffers passed to the "host-buffer" entry points are ordinary host memory
 * (the numpy arrays the reference passes to its Cython modules); the library
 * stages them through page-locked buffers OF ITS OWN and never hands a pageable
 * pointer to the HIP runtime (round 6: the runtime's transfers on pageable memory
 * returned wrong data with 16 or more processes per GPU).  Any
 * pointer may also be page-locked (mrchip_host_alloc): it then goes to the
 * runtime directly and the copy is a stream-ordered DMA.  The "page" entry points keep a page
 * resident on the device between the three yields of
 * mrc.create_mrc_hocr_components so that pixels cross PCIe once.
 *
 * Threading: a context owns its HIP streams and scratch; use one context per
 * thread (the reference is single-threaded and re-entrant).
 */
#ifndef MRCHIP_H
#define MRCHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRCHIP_ABI_VERSION 1

enum {
    MRCHIP_OK = 0,
    MRCHIP_E_HIP = -1,        /* HIP runtime error (see mrchip_last_error) */
    MRCHIP_E_ARG = -2,        /* invalid argument */
    MRCHIP_E_UNSUPPORTED = -3,/* parameter outside the implemented range */
    MRCHIP_E_NOMEM = -4,
    MRCHIP_E_STATE = -5       /* call out of order on a page handle */
};

typedef struct mrchip_ctx mrchip_ctx;
typedef struct mrchip_page mrchip_page;
typedef struct mrchip_batch mrchip_batch;

/* Row length of the per-page Gaussian weight tables handed to *_mask_finish
 * (2*radius+1 <= 121 entries used, radius <= 60). */
#define MRCHIP_MAX_TAPS 128

/* ---- lifecycle ---------------------------------------------------------- */
int mrchip_abi_version(void);
int mrchip_device_count(void);
/* One context per (thread, device).  NULL on failure. */
mrchip_ctx *mrchip_create(int device);
void mrchip_destroy(mrchip_ctx *ctx);
const char *mrchip_last_error(void);
int mrchip_sync(mrchip_ctx *ctx);
/* Debugging aid, no reference counterpart: with MRCHIP_CANARY=<KiB> in the environment every device block of the caching
 * allocator carries that many KiB of a fixed pattern on both sides (beyond the documented slack the kernels may touch).
 * This call waits for the device, verifies every guard band and reports on stderr; *bad_bytes = guard bytes found
 * overwritten since the context was created (always 0 with the switch off).  tests/fuzz_parity.py calls it per case. */
int mrchip_canary_check(mrchip_ctx *ctx, long long *bad_bytes);
/* Proof that the guards see a stray write: overwrites one byte on each side of a scratch block of its own (inside that
 * block's guard bands) and verifies; *detected = 2 with MRCHIP_CANARY on, 0 with it off. */
int mrchip_canary_selftest(mrchip_ctx *ctx, long long *detected);
/* Device name / CU count of the context's device (for reports). */
int mrchip_device_info(mrchip_ctx *ctx, char *name, int name_len, int *cus, size_t *hbm_bytes);
/* Device memory free / total right now (hipMemGetInfo), for callers that size batches: a batch of N pages holds about
 * (3 C + 9) W H N bytes of planes and scratch; allocations that would leave less than 2 GiB free are refused
 * (MRCHIP_E_NOMEM) instead of being tried. */
int mrchip_device_memory(mrchip_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
/* NUMA node of the host socket this context's GPU is attached to (sysfs), -1 if unknown: where a streaming caller
 * wants its page-locked buffers (mrchip_host_alloc from a thread running there). */
int mrchip_device_numa_node(mrchip_ctx *ctx);

/* Page-locked host memory for buffers handed to the *_async entry points (NULL on failure). */
void *mrchip_host_alloc(mrchip_ctx *ctx, size_t bytes);
void mrchip_host_free(mrchip_ctx *ctx, void *p);

/* ---- cython/sauvola.pyx -------------------------------------------------- */
/* sauvola.binarise_sauvola(in_arr, out_arr, width, height, window_width,
 * window_height, k, R) -- cython/sauvola.pyx:29-222.  in/out are flat
 * row-major uint8[w*h]; out = 1 for bright/background, 0 for dark (pyx:153).
 * invert != 0 stores the complement instead, i.e. mrc.threshold_image's
 * np.invert (mrc.py:85) fused.  Returns 0 like the reference (pyx:222). */
int mrchip_sauvola_u8(mrchip_ctx *ctx, const uint8_t *in, uint8_t *out, int w, int h,
                      int window_w, int window_h, double k, double R, int invert);

/* ---- cython/optimiser.pyx ------------------------------------------------ */
/* optimiser.fast_mask_denoise(mask, width, height, mincnt, n_size) --
 * cython/optimiser.pyx:436-472.  In place on uint8/bool[h][w]. */
int mrchip_mask_denoise(mrchip_ctx *ctx, uint8_t *mask, int w, int h, int mincnt, int n_size);

/* optimiser.optimise_gray2 / optimise_rgb2 (and the slow optimise_gray /
 * optimise_rgb, identical results) -- cython/optimiser.pyx:153-273, 280-429,
 * 22-146.  mask uint8/bool[h][w]; img/out uint8[h][w][channels] interleaved,
 * channels 1 or 3.  invert_mask != 0 uses (mask ^ 1), the reference's
 * mask_inv (mrc.py:439) fused. */
int mrchip_optimise(mrchip_ctx *ctx, const uint8_t *mask, const uint8_t *img, uint8_t *out,
                    int w, int h, int channels, int n_size, int invert_mask);

/* ---- internetarchivepdf/grayconvert.py:38-66 special_gray_convert (recode.py:362) ----
 * The reference computes per-channel min / max / mean / std (grayconvert.py:41-44), derives three level ranges from them
 * in scalar Python (:46-54), applies level_arr per channel (:24-31, :56-60), converts with skimage.color.rgb2hsv and
 * returns uint8(V (1 - S / 2) 255) (:62-66).  Here: `begin` uploads rgb uint8[h][w][3], keeps it on the device and returns
 * stats[12] = {min r g b, max r g b, sum r g b, sum of squares r g b} as exact integers; the host side (mrchip/grayconvert.py)
 * does the scalar arithmetic with the reference's own expressions and builds two byte tables -- level_luts[3][256]
 * (level_arr of every byte value, per channel) and hsl_table[256][256] (the rgb2hsv + lightness result of a pixel as a
 * function of its (max, min) after levelling, index max * 256 + min) -- which `finish` applies: out uint8[h][w].
 * `finish` consumes the pending page (MRCHIP_E_STATE without one); a second `begin` replaces it. */
int mrchip_special_gray_begin(mrchip_ctx *ctx, const uint8_t *rgb, int w, int h, unsigned long long *stats);
int mrchip_special_gray_finish(mrchip_ctx *ctx, const uint8_t *level_luts, const uint8_t *hsl_table, uint8_t *out);

/* ---- third-party stages reached from internetarchivepdf/mrc.py ----------- */
/* PIL Image.convert('L') of an RGB image -- mrc.py:361. */
int mrchip_luma601(mrchip_ctx *ctx, const uint8_t *rgb, uint8_t *gray, int w, int h);

/* mean_estimate_sigma(arr) = np.mean(skimage estimate_sigma(arr)) --
 * mrc.py:52-55.  kind 0: arr is a uint8 image taken as float32 values (the
 * grayimgf of mrc.py:372); kind 1: arr is a bool array (mrc.py:253-254,
 * PyWavelets' float64 path).  stride in elements. */
int mrchip_estimate_sigma(mrchip_ctx *ctx, const uint8_t *arr, int stride, int w, int h,
                          int kind, double *sigma);
/* mrc.estimate_noise(imgf) on float32(gray) -- mrc.py:273-296. */
int mrchip_estimate_noise_u8(mrchip_ctx *ctx, const uint8_t *gray, int w, int h, double *sigma);

/* scipy.ndimage.gaussian_filter(float32(gray), sigma).astype(uint8) --
 * mrc.py:311 + 325.  weights = the 2*radius+1 float64 table scipy builds on
 * the host (filters.py _gaussian_kernel1d); NULL = build it here with libm. */
int mrchip_gaussian_u8(mrchip_ctx *ctx, const uint8_t *gray, uint8_t *out, int w, int h,
                       double sigma, const double *weights, int radius);
/* The same stages for a float32 image that does NOT hold whole numbers 0..255 (mrc.estimate_noise / create_threshold_mask,
 * mrc.py:273-329, take any float32 array; the production path, mrc.py:372, passes float32(gray) and uses the uint8 entry
 * points).  estimate_sigma_f32: `stride` in elements.  gaussian_f32: out = uint8(float32 result) -- scipy's float32
 * gaussian_filter followed by the astype(np.uint8) of mrc.py:325 (truncation; the caller guarantees values in [0, 256),
 * outside of which numpy's own cast is platform-defined); radius 0 with weights NULL = no blur, the cast alone. */
int mrchip_estimate_sigma_f32(mrchip_ctx *ctx, const float *arr, int stride, int w, int h, double *sigma);
int mrchip_estimate_noise_f32(mrchip_ctx *ctx, const float *gray, int w, int h, double *sigma);
int mrchip_gaussian_f32(mrchip_ctx *ctx, const float *in, uint8_t *out, int w, int h, double sigma,
                        const double *weights, int radius);

/* PIL Image.thumbnail((req_w, req_h)) with defaults (BICUBIC, reducing_gap=2)
 * -- mrc.py:422-428, 456-462.  mrchip_thumbnail_size is the host-side size
 * rule (returns 1 if the image changes, 0 if left untouched). */
int mrchip_thumbnail_size(int w, int h, int req_w, int req_h, int *out_w, int *out_h);
int mrchip_thumbnail(mrchip_ctx *ctx, const uint8_t *in, int w, int h, int channels,
                     int req_w, int req_h, uint8_t *out /* out_w*out_h*channels */);
/* PIL Image.thumbnail((req_w, req_h), resample=filter, reducing_gap=gap) -- the page-ingest
 * downsample of recode.py:368-372 is (LANCZOS, None); reducing_gap <= 0 stands for None (no
 * Image.reduce step).  Same size rule as above (req_* = floor of the float size PIL is given). */
enum { MRCHIP_FILTER_BICUBIC = 0, MRCHIP_FILTER_LANCZOS = 1 };
int mrchip_thumbnail_ex(mrchip_ctx *ctx, const uint8_t *in, int w, int h, int channels,
                        int req_w, int req_h, int filter, double reducing_gap, uint8_t *out);

/* ---- internetarchivepdf/mrc.py ------------------------------------------- */
/* mrc.threshold_image window rule -- mrc.py:68-75. */
int mrchip_window_for_dpi(int has_dpi, double dpi);

/* mrc.create_hocr_mask pixel work -- mrc.py:223-266 -- for boxes that passed
 * the text/confidence/geometry filter (mrc.py:198-221, host logic).
 * boxes = nb x {left, top, right, bottom}, committed in list order.
 * decisions (optional, nb): 0 none, 1 thres, 2 thres_invert. */
int mrchip_hocr_mask(mrchip_ctx *ctx, const uint8_t *gray, uint8_t *mask, int w, int h,
                     const int32_t *boxes, int nb, int window, int32_t *decisions);

/* ---- device-resident page: mrc.create_mrc_hocr_components (mrc.py:334-471) */
mrchip_page *mrchip_page_create(mrchip_ctx *ctx, int w, int h, int channels);
void mrchip_page_destroy(mrchip_page *pg);
/* uint8[h][w][channels] host image -> device (asynchronous on the page's stream) */
int mrchip_page_upload(mrchip_page *pg, const uint8_t *img);
/* see mrchip_batch_upload_gray / mrchip_batch_upload_mask */
int mrchip_page_upload_gray(mrchip_page *pg, const uint8_t *gray);
int mrchip_page_upload_mask(mrchip_page *pg, const uint8_t *mask);
/* Phase A (enqueue only): luma, both hOCR-box thresholds + counts, noise estimate. */
int mrchip_page_mask_begin(mrchip_page *pg, const int32_t *boxes, int nb, int window);
/* Waits for phase A; returns sigma_est (mrc.py:305).  The caller (host, like
 * scipy) builds the Gaussian table for sigma_est*0.1 when sigma_est > 1. */
int mrchip_page_sigma(mrchip_page *pg, double *sigma_est);
/* Phase B (enqueue only): box decisions + commit, blur, Sauvola k=0.34, OR,
 * fast denoise.  weights/radius as in mrchip_gaussian_u8 (ignored unless
 * sigma_est > 1). */
int mrchip_page_mask_finish(mrchip_page *pg, const double *weights, int radius, int denoise_fast);
/* first yield: bool[h][w] */
int mrchip_page_download_mask(mrchip_page *pg, uint8_t *mask);
/* The same mask at 1 bit per pixel, most significant bit first, rows of (w+7)/8 bytes -- the raw
 * PBM (P4) / PIL mode '1' layout mrc.encode_mrc_mask builds before jbig2/PNG (mrc.py:474-520;
 * SURVEY.md 8f rank 1).  An eighth of the bytes cross PCIe. */
int mrchip_page_download_mask_packed(mrchip_page *pg, uint8_t *packed);
/* second / third yield (enqueue only): optimise (fg: n=3 on mask; bg: n=10 on
 * the inverted mask) + optional thumbnail.  downsample <= 0: none.
 * *too_small = 1 reproduces 'too-small-to-downsample' (mrc.py:429-431). */
int mrchip_page_layer(mrchip_page *pg, int is_bg, double downsample, int *out_w, int *out_h,
                      int *too_small);
/* both layers in one launch (the generator's second yield: a single page is latency-bound per launch, so fg and bg run
 * side by side; the third yield then only downloads).  too_small: bit 0 fg, bit 1 bg. */
int mrchip_page_layers(mrchip_page *pg, double fg_downsample, double bg_downsample, int *fg_w, int *fg_h,
                       int *bg_w, int *bg_h, int *too_small);
int mrchip_page_download_layer(mrchip_page *pg, int is_bg, uint8_t *out);
int mrchip_page_sync(mrchip_page *pg);
/* decisions of the hOCR boxes after mask_finish (diagnostics / tests) */
int mrchip_page_box_decisions(mrchip_page *pg, int32_t *decisions, int nb);
/* device pointers for zero-copy consumers (mask [h][pitch], layer tight) */
int mrchip_page_device_ptrs(mrchip_page *pg, void **img, void **mask, size_t *mask_pitch,
                            void **fg, void **bg);

/* ---- page batches: N same-sized pages, every stage one launch over the batch ---
 * The unit recode.py's page loop (recode.py:291) would hand over when it
 * decomposes several pages at once; the single-page handle above is a batch of
 * one.  Calls mirror the page calls; `page` indexes the batch. */
mrchip_batch *mrchip_batch_create(mrchip_ctx *ctx, int npages, int w, int h, int channels);
void mrchip_batch_destroy(mrchip_batch *b);
int mrchip_batch_upload(mrchip_batch *b, int page, const uint8_t *img);
/* Gray plane of an RGB page supplied by the caller instead of the Rec.601 luma of its pixels:
 * create_mrc_hocr_components thresholds `image.convert('L')` of the ORIGINAL image (mrc.py:359-361) and
 * converts modes other than L / RGB to RGB only for the layers (mrc.py:401-404).  After mrchip_batch_upload. */
int mrchip_batch_upload_gray(mrchip_batch *b, int page, const uint8_t *gray);
/* Replace the finished mask of a page (uint8/bool[h][w]) before mrchip_batch_layers: the hook for mask
 * post-processing kept on the host -- denoise_mask='bregman' (mrc.py:90-108, 391-392). */
int mrchip_batch_upload_mask(mrchip_batch *b, int page, const uint8_t *mask);
/* Number of pages in use, 1..npages (default npages): one batch object serves a short last batch of a
 * stream; pages >= count are neither read nor written by any stage. */
int mrchip_batch_set_count(mrchip_batch *b, int count);
/* filtered hOCR line boxes of one page (mrc.py:198-221 is host logic), list order */
int mrchip_batch_set_boxes(mrchip_batch *b, int page, const int32_t *boxes, int nb);
int mrchip_batch_mask_begin(mrchip_batch *b, int window);
/* mrc.threshold_image(gray, dpi, k) (mrc.py:58-87) of every page in one launch (window from
 * mrchip_window_for_dpi): True = dark into the mask plane, downloaded with mrchip_batch_download_mask*.
 * Enqueue only. */
int mrchip_batch_threshold(mrchip_batch *b, int window, double k);
/* waits for phase A; sigma_est[npages] (mrc.py:305) */
int mrchip_batch_sigmas(mrchip_batch *b, double *sigma_est);
/* weights: npages rows of MRCHIP_MAX_TAPS doubles (row i used iff sigma_est[i] > 1),
 * radius[npages]; both NULL = build the tables here with libm's exp */
int mrchip_batch_mask_finish(mrchip_batch *b, const double *weights, const int *radius, int denoise_fast);
int mrchip_batch_download_mask(mrchip_batch *b, int page, uint8_t *mask);
int mrchip_batch_download_mask_packed(mrchip_batch *b, int page, uint8_t *packed);
/* which: 1 = fg, 2 = bg, 3 = both (one launch).  downsample <= 0: none.
 * too_small: bit 0 fg, bit 1 bg ('too-small-to-downsample', mrc.py:429-431, 463-465) */
int mrchip_batch_layers(mrchip_batch *b, int which, double fg_downsample, double bg_downsample,
                        int *fg_w, int *fg_h, int *bg_w, int *bg_h, int *too_small);
int mrchip_batch_download_layer(mrchip_batch *b, int page, int is_bg, uint8_t *out);
/* fg/bg hand-off to the image encoders (mrc.py:523-580 writes each layer to a file for kdu/opj/grok;
 * SURVEY.md 8f rank 2): enqueue-only copy on the batch's stream when `out` is page-locked (mrchip_host_alloc) -- a
 * true asynchronous DMA -- so page i can go to its encoder while page i+1 is still on the device; mrchip_batch_sync
 * before the bytes are read.  With `out` in ordinary memory the call is correct but not asynchronous: it returns when
 * the bytes are in place (staged through the library's page-locked buffers). */
int mrchip_batch_download_layer_async(mrchip_batch *b, int page, int is_bg, uint8_t *out);
/* the same for the mask (bool bytes / 1 bpp): what a streaming caller queues behind mrchip_batch_layers so
 * that batch i-1 leaves over PCIe while batch i is computed and batch i+1 arrives (recode.py:291's page
 * loop, pipelined) */
int mrchip_batch_download_mask_async(mrchip_batch *b, int page, uint8_t *mask);
int mrchip_batch_download_mask_packed_async(mrchip_batch *b, int page, uint8_t *packed);
/* 1 when everything queued on the batch's stream has finished, 0 while work is pending; never blocks */
int mrchip_batch_done(mrchip_batch *b);
int mrchip_batch_sync(mrchip_batch *b);
int mrchip_batch_box_decisions(mrchip_batch *b, int page, int32_t *decisions, int nb);
int mrchip_batch_device_ptrs(mrchip_batch *b, int page, void **img, void **mask, size_t *mask_pitch,
                             void **fg, void **bg);

/* Device self-test of the exact fp64 quotient behind mrchip_sauvola_u8 (cython/sauvola.pyx:144-145 are
 * truncating integer divisions): every divisor 1..65792, the dividends around each multiple.
 * *mismatches = number of (dividend, divisor) pairs whose quotient differs from integer division. */
int mrchip_selftest_sauvola_quotients(mrchip_ctx *ctx, long long *mismatches);
/* Device self-test of the table-driven decision of mrchip_sauvola_u8 for one (k >= 0, R) (cython/sauvola.pyx:62,
 * 143-153): for every integer mean 0..255, pixel 0..255 and variance 0..255 mean + 254 - mean^2 (every value a
 * window of 8-bit pixels can produce) the table's answer `variance + mean^2 >= T2[mean][pixel]` against the
 * reference's fp64 sequence.  *tested = triples compared, *table_bytes = size of the table (NULL: not wanted).
 * MRCHIP_E_UNSUPPORTED if this (k, R) has no table (the launch then uses the fp64 sequence for every pixel). */
int mrchip_selftest_sauvola_table(mrchip_ctx *ctx, double k, double R, long long *mismatches, long long *tested,
                                  int *table_bytes);
/* Same for mrchip_optimise's `val / cnt` (cython/optimiser.pyx:261-269): every count 1..5120 (n_size <= 32)
 * against every value 0..255*count. */
int mrchip_selftest_optimise_quotients(mrchip_ctx *ctx, long long *mismatches);
/* Device self-test of the float32 form of the Gaussian pre-blur (scipy.ndimage.gaussian_filter at mrc.py:309-311): for a
 * weights table of radius 1 or 2 (2 * radius + 1 doubles, as scipy builds it), every (centre byte, pair sum, pair sum)
 * through the kernel's float32 vertical sum and through the reference's float64 one; *mismatches = the number of triples
 * whose difference exceeds the bound the undecided-pixel test is derived from (must be 0), *max_error = the largest
 * difference seen. */
int mrchip_selftest_gauss_fast(mrchip_ctx *ctx, const double *weights, int radius, long long *mismatches, double *max_error);

/* ---- page sharding across GPUs: control-plane collectives over RCCL ---------------------------------------
 * Pages are independent (recode.py:291's loop body reads only page idx): one process per GPU, page i -> rank
 * i mod G, no pixel ever crosses ranks.  What does cross is bytes of control data -- the page descriptor table
 * and flattened hOCR boxes from rank 0, per-page result records back, the maximum of the elapsed times -- and it
 * goes through these calls (librccl.so, opened at run time).  Rank 0 makes the id with mrchip_comm_unique_id and
 * hands its 128 bytes to the other ranks by any side channel (mrchip/dist.py uses a file). */
typedef struct mrchip_comm mrchip_comm;
int mrchip_comm_unique_id(unsigned char *id128);
mrchip_comm *mrchip_comm_init(mrchip_ctx *ctx, int rank, int world, const unsigned char *id128);
void mrchip_comm_destroy(mrchip_comm *c);
/* `bytes` of host memory from rank `root` to every rank, in place */
int mrchip_comm_bcast(mrchip_comm *c, void *buf, size_t bytes, int root);
/* every rank contributes `bytes`; recv[world * bytes] holds the contributions in rank order on every rank */
int mrchip_comm_allgather(mrchip_comm *c, const void *send, size_t bytes, void *recv);
/* op 0: sum, 1: max over ranks of n doubles, in place (n = 1 doubles as the barrier) */
int mrchip_comm_allreduce_f64(mrchip_comm *c, double *vals, int n, int op);

/* ---- measurement --------------------------------------------------------- */
/* Per-kernel HIP-event timing on the stream each kernel is launched on.
 * enable: 0 off, 1 on.  Kernels are named as in the kernel trace. */
int mrchip_prof_enable(mrchip_ctx *ctx, int enable);
int mrchip_prof_reset(mrchip_ctx *ctx);
/* Resolves pending events (synchronises).  Returns the number of distinct
 * kernels; index i in [0,n): name, launches, total milliseconds,
 * algorithmic bytes accumulated over those launches. */
int mrchip_prof_count(mrchip_ctx *ctx);
int mrchip_prof_get(mrchip_ctx *ctx, int i, char *name, int name_len, long long *launches,
                    double *total_ms, double *alg_bytes);
/* Measured HBM ceiling to quote next to the 8 TB/s spec peak: device-to-device
 * copy of `bytes` repeated `reps` times, timed with HIP events; *gbps = bytes read + written
 * per second / 1e9. */
int mrchip_hbm_copy_bandwidth(mrchip_ctx *ctx, size_t bytes, int reps, double *gbps);

#ifdef __cplusplus
}
#endif
#endif /* MRCHIP_H */
