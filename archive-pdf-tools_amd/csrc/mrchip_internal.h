// Internal declarations shared by the libmrchip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mrchip.h"

#define MRCHIP_EXPORT extern "C" __attribute__((visibility("default")))

namespace mrchip {

void set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            mrchip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
                              hipGetErrorString(_e));                                    \
            return MRCHIP_E_HIP;                                                         \
        }                                                                                \
    } while (0)

#define TRY(expr)                      \
    do {                               \
        int _r = (expr);               \
        if (_r != 0) return _r;        \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope
// fence for GLOBAL memory, i.e. `s_waitcnt vmcnt(0)`: in the row-sequential kernels that would expose
// the full latency of the prefetched global loads and of the row's stores on every row.  These
// kernels never exchange data between lanes through global memory, so only lgkmcnt must drain.
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// same for a single-wave workgroup: LDS operations of one wave execute in order, only the
// compiler and the lgkm counter need to be told
__device__ __forceinline__ void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif

// All device images are pitched (pitch % 64 == 0) with PAD bytes of slack in
// front of row 0 and behind the last row, so kernels may issue aligned dword
// loads that straddle the row ends (the bytes are masked, never used).
constexpr int PAD = 1024;

struct DevBlock {
    void *base = nullptr;   // hipMalloc'ed
    size_t bytes = 0;       // of the whole block, guard bands included
    bool busy = false;
    size_t guard = 0;       // MRCHIP_CANARY: bytes of pattern on each side; the caller's pointer is base + guard
};

struct ProfEntry {
    std::string name;
    long long launches = 0;
    double ms = 0;
    double alg_bytes = 0;
};

struct ProfPending {
    int entry;
    hipEvent_t a, b;
};

struct OptMail;
struct GrayPending;
constexpr int NSTREAMS = 4;
constexpr int MAX_GRID_Z = 65535;      // hipDeviceProp_t::maxGridSize[2] (and [1])

}  // namespace mrchip

struct mrchip_ctx {
    int device = 0;
    hipStream_t streams[mrchip::NSTREAMS] = {};
    int next_stream = 0;
    std::vector<mrchip::DevBlock> blocks;
    // pinned staging for small control data (descriptors, results)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    // profiling
    bool prof = false;
    std::vector<mrchip::ProfEntry> prof_entries;
    std::vector<mrchip::ProfPending> prof_pending;
    std::vector<hipEvent_t> event_pool;
    // while an entry point that enqueues asynchronous work on scratch buffers is running: the stream
    // its DevBufs must wait for before they go back to the allocator (error paths return early)
    hipStream_t scratch_sync = nullptr;
    // hand-off buffers / error word of mrchip_optimise's launches (one per context: a per-call object would allocate and free
    // page-locked memory -- two device synchronisations -- on every call)
    mrchip::OptMail *host_mail = nullptr;
    // the page mrchip_special_gray_begin left on the device for mrchip_special_gray_finish
    mrchip::GrayPending *gray_pending = nullptr;
    int cus = 0;
    size_t hbm = 0;
    char name[128] = {};
    long long canary_bad = 0;     // guard-band bytes found overwritten so far (MRCHIP_CANARY)
};

namespace mrchip {

// caching device allocator (host-side bookkeeping only)
int dev_alloc(mrchip_ctx *ctx, size_t bytes, void **out);
void dev_free(mrchip_ctx *ctx, void *p);

struct DevBuf {   // RAII for scratch inside an entry point
    mrchip_ctx *ctx = nullptr;
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(mrchip_ctx *c, size_t bytes) {
        release();
        ctx = c;
        return dev_alloc(c, bytes, &p);
    }
    void release() {
        if (p) {
            // an early (error) return may leave kernels / copies in flight on the entry point's stream
            if (ctx->scratch_sync) (void)hipStreamSynchronize(ctx->scratch_sync);
            dev_free(ctx, p);
        }
        p = nullptr;
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

// Declared first in a host-buffer entry point: every scratch DevBuf released while it is alive waits for
// `s` (free on the success path, where the stream is already idle).
struct ScratchSync {
    mrchip_ctx *ctx;
    hipStream_t prev;
    ScratchSync(mrchip_ctx *c, hipStream_t s) : ctx(c), prev(c->scratch_sync) { c->scratch_sync = s; }
    ~ScratchSync() { ctx->scratch_sync = prev; }
    ScratchSync(const ScratchSync &) = delete;
    ScratchSync &operator=(const ScratchSync &) = delete;
};

// pitched 8-bit image (c interleaved channels per pixel)
struct Img8 {
    DevBuf buf;
    uint8_t *p = nullptr;  // row 0
    int w = 0, h = 0, c = 1, pitch = 0;
    int alloc(mrchip_ctx *ctx, int w_, int h_, int c_ = 1) {
        w = w_; h = h_; c = c_;
        pitch = round_up(w * c + 64, 64);
        TRY(buf.alloc(ctx, (size_t)pitch * h + 2 * PAD));
        p = buf.as<uint8_t>() + PAD;
        return 0;
    }
    size_t bytes() const { return (size_t)pitch * h; }
};

// grayconvert.py: the RGB page between the statistics pass and the table pass (it crosses PCIe once)
struct GrayPending {
    Img8 src;
    int w = 0, h = 0;
};
struct RgbStats {                     // what the statistics kernel leaves for the host (80 bytes)
    unsigned mn[3], mx[3];
    unsigned pad_[2];
    unsigned long long sum[3], sumsq[3];
};
// d_stats: one RgbStats
int launch_rgb_stats(mrchip_ctx *ctx, hipStream_t s, const uint8_t *rgb, int pitch, int w, int h, void *d_stats);
// d_tables: 3 * 256 bytes of per-channel level tables, then 256 * 256 bytes indexed max * 256 + min
int launch_rgb_level_hsl(mrchip_ctx *ctx, hipStream_t s, const uint8_t *rgb, int rgb_pitch, uint8_t *gray, int gray_pitch, int w,
                         int h, const uint8_t *d_tables);

int upload_2d(hipStream_t s, uint8_t *dst, int dpitch, const uint8_t *src, int spitch, int row_bytes, int rows);
int upload_1d(hipStream_t s, void *dst, const void *src, size_t bytes);
int download_2d(hipStream_t s, uint8_t *dst, int dpitch, const uint8_t *src, int spitch, int row_bytes, int rows);
// Transfers whose host side is pageable go through page-locked staging buffers of the library (ctx.hip); a download into
// pageable memory returns with the data in place, everything else is stream-ordered and asynchronous
int download_1d(hipStream_t s, void *dst, const void *src, size_t bytes);

// profiling hooks: prof_begin returns a token (<0: disabled)
int prof_begin(mrchip_ctx *ctx, hipStream_t s, const char *name, double alg_bytes);
void prof_end(mrchip_ctx *ctx, hipStream_t s, int token);
int prof_resolve(mrchip_ctx *ctx);

#define LAUNCH(ctx, stream, name, alg_bytes, ...)                                   \
    do {                                                                            \
        int _tok = mrchip::prof_begin(ctx, stream, name, (double)(alg_bytes));      \
        __VA_ARGS__;                                                                \
        mrchip::prof_end(ctx, stream, _tok);                                        \
        HIP_TRY(hipGetLastError());                                                 \
    } while (0)

// ---- stage launchers (device pointers, asynchronous on `s`) -----------------

// One Sauvola job = one image or crop treated as a standalone image.
struct SauvolaJob {
    const uint8_t *src;   // pixel (0,0) of the crop
    int src_pitch;
    int w, h;
    uint8_t *dst;         // threshold output for the crop's pixel (0,0) (polarity A)
    int dst_pitch;
    uint8_t *dst_inv;     // optional: output for the 255-p image (polarity B), same pitch
    unsigned int *counts; // optional: counts[0] += #ones(A), counts[1] += #ones(B)
    // optional (page jobs on the 8- / 16-column kernels, see sauvola_writes_bits): polarity A also at 1 bit per pixel,
    // bit k of byte j of a row = column 8 j + k -- the little-endian 32-pixel words of the denoiser's rows.  Bytes past
    // ceil(w / 8) of a row are never written.
    uint8_t *bits;
    int bits_pitch;       // bytes per row of the bit plane
    // with bits: the caller does not read `dst` before it rewrites it (the batch's mask bytes are unpacked from the
    // denoiser's bit rows afterwards): the 8-column page kernel then stores the bit rows only
    int no_bytes;
};
// true when launch_sauvola_dev will take a kernel that can fill SauvolaJob::bits for jobs of this size / window
bool sauvola_writes_bits(int maxw, int maxh, int ww);

enum SauvolaFlags {
    SAUVOLA_INVERT = 1,   // store 1 for dark (mrc.threshold_image polarity)
};

int launch_sauvola(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *jobs, int njobs,
                   int ww, int wh, double k, double R, int flags);

// a plane of a page batch: page i, row y starts at p + i*stride + y*pitch
struct Plane {
    uint8_t *p = nullptr;
    int pitch = 0;
    size_t stride = 0;
    uint8_t *page(int i) const { return p + (size_t)i * stride; }
};

int launch_luma601(mrchip_ctx *ctx, hipStream_t s, Plane rgb, Plane gray, int w, int h, int npages);


struct OptJob {
    const uint8_t *mask; int mpitch;
    const uint8_t *img; int ipitch;
    uint8_t *out; int opitch;
    int w, h, n, invert;
    // optional: the same mask at 1 bit per pixel (LSB first, mwpr dwords per row) as the denoiser leaves it;
    // the packed kernel reads it instead of the byte mask (an eighth of the traffic of 3 reads per row)
    const unsigned *mbits; int mwpr;
    // optional, with mbits: one byte per row, 1 = the row of the (uninverted) bit mask has a set pixel (the denoiser leaves
    // them); saves the band scan its pass over the bit rows
    const uint8_t *rowflags;
    // band walkers only (launch_optimise_jobs clears it otherwise): do not copy the rows outside the bands into `out`;
    // `rowmap` (set by the launcher, cdiv(h, 32) words) gets bit y = 1 for the rows that were written
    int skip_copy;
    unsigned *rowmap;
};
// hand-off buffer of the strip schedules (one per owner that may have a launch in flight: a batch, or a host-buffer
// call).  The first 256 bytes of `buf` are unused padding; `err` is a page-locked host word the kernels set when a
// bounded poll gives up (the owner reads it after it has synchronised with the stream: optmail_check).
struct OptMail {
    DevBuf buf;
    size_t bytes = 0;
    unsigned epoch = 0;
    unsigned *err = nullptr;
    DevBuf bits;              // 1-bpp copy of a byte mask for callers that have none (host-buffer entry point)
    size_t bits_bytes = 0;
    DevBuf bandq;             // queue of the band walkers: 256 bytes of control words + OptBand entries
    size_t bandq_bytes = 0;
    OptMail() = default;
    OptMail(const OptMail &) = delete;
    OptMail &operator=(const OptMail &) = delete;
    ~OptMail();
};
// MRCHIP_E_HIP (and the word cleared) if a launch since the last check reported a hand-off timeout; call with the
// stream idle
int optmail_check(OptMail *mail);
// h_jobs: host copy of the job records (the launcher fills the schedule's fields and uploads them to d_jobs on `s`;
// it must stay valid until the copy has run)
int launch_optimise_jobs(mrchip_ctx *ctx, hipStream_t s, OptJob *h_jobs, OptJob *d_jobs, int njobs, int w, int h, int c,
                         int n_max, OptMail *mail);
// bytes != 0 -> 1 bit per pixel, LSB first, wpr dwords per row (the denoiser's rows)
int launch_pack_bits(mrchip_ctx *ctx, hipStream_t s, const uint8_t *mask, int pitch, int w, int h, unsigned *bits, int wpr);


// hOCR: commit chosen thresholds into the mask in list order
struct HocrBox {
    int l, t, r, b;
    int decision;            // 0 none, 1 thres, 2 thres_invert
    const uint8_t *th;       // polarity A scratch (pixel (0,0) of the box)
    const uint8_t *thi;      // polarity B scratch
    int pitch;
    uint8_t *mask;           // row 0 of the page's mask
    int mpitch;
    int page_end;            // index one past the last box of the same page
    int overlapped;          // a later box of the page with a decision intersects this one
    uint8_t *bits;           // optional (or_mode): row 0 of the page's 1-bpp mask rows, OR-ed alongside the bytes
    int bits_pitch;
    int no_bytes;            // with bits: only the bit rows are updated (the mask bytes are rewritten from them afterwards)
};
int launch_hocr_commit(mrchip_ctx *ctx, hipStream_t s, const HocrBox *d_boxes, int nb, int maxw, int maxh, double area,
                       int or_mode);

constexpr int THUMB_MAXK = 20;
// thumbnail plan: host-side size rule + fixed-point coefficient tables
struct ThumbPlan {
    int w = 0, h = 0, c = 1;
    int filter = 0;  double reducing_gap = 2.0;     // Image.thumbnail's resample / reducing_gap
    int changed = 0;          // 0: image left untouched
    int ow = 0, oh = 0;       // output size
    int fx = 1, fy = 1;       // Image.reduce factors
    int rw = 0, rh = 0;       // size after reduce
    int need_h = 0, need_v = 0, ksh = 0, ksv = 0;
    std::vector<int32_t> bh_, kh_, bv_, kv_, khT_;   // khT_: kh_ transposed, padded to THUMB_MAXK taps
    // Matrix-core path: a resize pass is the product of the pixel lines with a banded coefficient
    // matrix.  Per tile of 16 outputs: first input byte (kbase), per output the constant term
    // (bias), and the MFMA B operand (KB blocks of 64 input bytes x 3 balanced base-256 digits of
    // the 22-bit coefficients, already in lane order).
    struct Mm {
        int ntiles = 0, KB = 0, nout = 0, kalign = 1, panel_w = 0, panel_w8 = 0;   // panel_w / panel_w8: input bytes spanned by 16 / 8 consecutive tiles
        std::vector<int32_t> kbase, kend, bias; std::vector<unsigned char> b;     // kend: one past the last input byte a tile has a tap on (running maximum)
    };
    Mm mmh, mmv;
    int fuse_rv = 0;          // > 0: both passes in one kernel, this many vertical tiles per workgroup
    int mm_ok = 0;            // both passes present and each 16-output tile spans <= 128 input bytes
    // every table in one blob (what the device copy holds), byte offsets of the parts
    std::vector<unsigned char> blob_;
    size_t off_bh = 0, off_kh = 0, off_bv = 0, off_kv = 0, off_khT = 0, off_mm[2][3] = {{0, 0, 0}, {0, 0, 0}}, off_mmend[2] = {0, 0};
};
int ThumbPlan_build(ThumbPlan &p, int w, int h, int c, int req_w, int req_h, int filter = 0 /* MRCHIP_FILTER_BICUBIC */,
                    double reducing_gap = 2.0);
size_t ThumbPlan_table_bytes(const ThumbPlan &p);          // == p.blob_.size(): copy p.blob_ to the device
// per-page extent of the pass-to-pass scratch image (row-major ow*c x rh, or transposed + padded for the matrix-core path)
void ThumbPlan_scratch2_dims(const ThumbPlan &p, int *width_bytes, int *rows);
// dst: page i at dst + i*dstride (tight rows of dpitch bytes); scratch1/2 likewise with their strides
// Rows of the source that live in another plane: row y of page i comes from `src` where bit y of rowmap[i * cdiv(h, 32) ..]
// is set, from `alt` where it is clear (one-kernel matrix-core form only: thumbnail_reads_source_fused)
struct ThumbAlt { Plane alt; const unsigned *rowmap; };
bool thumbnail_reads_source_fused(const ThumbPlan &p, Plane src, Plane dst);
int launch_thumbnail_plan(mrchip_ctx *ctx, hipStream_t s, const ThumbPlan &p, Plane src, Plane dst,
                          const void *d_tables, Plane scratch1, Plane scratch2, int npages, const ThumbAlt *ta = nullptr);
size_t sigma_scratch_bytes(int w, int h, int kind);
// one noise estimate = one job (a page's central crop, or one bool threshold of an hOCR box)
struct SigJob {
    const uint8_t *src; int pitch;
    int w, h;
    int as_bool;        // (the launch's `kind` decides how src is read: 0 uint8 as float32, 1 bool as float64, 2 float32 plane)
    char *scratch;      // sigma_scratch_bytes(w, h, kind) bytes, 256-byte aligned
};
int launch_estimate_sigma_jobs(mrchip_ctx *ctx, hipStream_t s, const SigJob *h_jobs, const SigJob *d_jobs, int njobs,
                               int kind, double *d_sigma);
constexpr int GMAXR = 60;
struct GaussW { double w[2 * GMAXR + 1]; int radius; int pad_; };
// d_weights: npages GaussW records (radius 0 = identity); tmp: page i at tmp + i*tstride floats
bool gauss_uses_fused(int w, int h, int max_radius);
int gauss_fast_selftest(mrchip_ctx *ctx, hipStream_t s, const GaussW *d_w, unsigned long long *d_bad, unsigned *d_maxerr);
void gauss_pad_weights(GaussW &g, int R);
// max_radius: the largest radius among the pages (host knows it): <= 8 takes the fused LDS kernel,
// which expects every page's table padded to max_radius (gauss_pad_weights)
int launch_gaussian_f32(mrchip_ctx *ctx, hipStream_t s, const float *src, int spitch, Plane dst, int w, int h,
                        const GaussW *d_weights, float *tmp, int tpitch);
// fast_ok: every page's table passed gauss_weights_allow_fast (before padding), so the float32 form may be taken
bool gauss_weights_allow_fast(const GaussW &g);
int launch_gaussian_batch(mrchip_ctx *ctx, hipStream_t s, Plane src, Plane dst, int w, int h, const GaussW *d_weights,
                          float *tmp, int tpitch, size_t tstride, int npages, int max_radius, bool fast_ok = true);
// bits: page i at bits + i*bits_stride dwords
// want_rowflags: also leave one byte per row ("the finished row has a set pixel") at denoise_rowflags_offset(w, h) of
// each page's scratch (n = 2, mincnt = 4 path only)
int launch_denoise_batch(mrchip_ctx *ctx, hipStream_t s, Plane mask, int w, int h, int mincnt, int n, unsigned *bits,
                         size_t bits_stride, int npages, bool bits_ready = false, bool want_rowflags = false);
size_t denoise_scratch_bytes(int w, int h);
size_t denoise_rowflags_offset(int w, int h);
int sauvola_div_selftest(mrchip_ctx *ctx, hipStream_t s, unsigned long long *d_bad);
// d_bad_tested[0] += mismatching, [1] += tested (mean, px, var) triples of the decision table of (k, R)
int sauvola_table_selftest(mrchip_ctx *ctx, hipStream_t s, double k, double R, unsigned long long *d_bad_tested, int *table_bytes);
int optimise_div_selftest(mrchip_ctx *ctx, hipStream_t s, unsigned long long *d_bad);
// 1 bpp MSB-first rows of (w+7)/8 bytes; page i at out + i*ostride
int launch_pack_msb(mrchip_ctx *ctx, hipStream_t s, Plane mask, int w, int h, uint8_t *out, size_t ostride, int npages);

// host logic
int thumbnail_size(int w, int h, int req_w, int req_h, int *ow, int *oh);
int gaussian_weights_libm(double sigma, std::vector<double> &w);

}  // namespace mrchip

namespace mrchip {
int launch_sauvola_dev(mrchip_ctx *ctx, hipStream_t s, const SauvolaJob *jobs, const SauvolaJob *d_jobs,
                       int njobs, int ww, int wh, double k, double R, int flags);
}
