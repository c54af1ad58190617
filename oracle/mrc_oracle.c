/*
 * mrc_oracle.c -- CPU restatement of the reference MRC page-decomposition hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed CPU
 * baseline ("port").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (archive-pdf-tools_amd/) never
 * links, imports or calls anything in oracle/; it fails loudly when the HIP
 * library is missing.
 *
 * Parity pinning: the reference ships no tests for this path, so
 * every function here is pinned against the REAL reference run in the build
 * container (oracle/ref_loader.py imports internetarchivepdf/mrc.py and the
 * Cython modules compiled from /root/reference by oracle/build_ref.sh):
 *   - committed golden vectors tests/golden/ (npz files, digests.json) made by
 *     tests/golden/make_golden.py from the reference itself, which also asserts the
 *     SURVEY.md 8c known-answer digests while it runs (build container only);
 *   - tests/test_oracle_golden.py compares this file with those vectors anywhere.
 *
 * Each function cites the reference file:line it restates.  Third-party
 * algorithms that the reference reaches through un-vendored dependencies
 * (Pillow 11.3.0 pinned in requirements.txt:1; scipy>=1.7.2; scikit-image
 * >=0.18.3 -> PyWavelets) are restated from their published algorithms and
 * pinned by the same goldens (probed here with Pillow 8.4.0/12.2.0, scipy
 * 1.7.1, scikit-image 0.18.3, PyWavelets 1.1.1).
 *
 * Plain C99, no dependencies.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------- */
/* a1: sauvola.binarise_sauvola  (cython/sauvola.pyx:29-222)                  */
/*                                                                           */
/* Closed form of the streaming loops: for output (y,x) the window is        */
/*   rows (y-o, y+u] n [0,H)   with o=(wh+1)/2, u=wh/2   (sauvola.pyx:78-79) */
/*   cols (x-l, x+r] n [0,W)   with l=(ww+1)/2, r=ww/2   (sauvola.pyx:76-77) */
/* S = sum px, Q = sum px^2, count = clipped area.                           */
/*   mean = (double)(S / count)            C integer division (cdivision)    */
/*   variance = (double)(Q / count) - mean*mean                              */
/*   tmp = px + mean*(k-1)                                                   */
/*   k>=0: form = tmp<=0 || tmp*tmp <= mean*mean*k2*variance  (pyx:143-147)  */
/*   k<0 : form = tmp<=0 && tmp*tmp >= mean*mean*k2*variance  (pyx:148-152)  */
/*   out = form ? 0 : 1                                        (pyx:153)     */
/* k2 = k*k/R/R (pyx:62).  Column sums are int32 like the reference's        */
/* `integral`/`integral_square` arrays (pyx:64-65), the window sum of        */
/* squares is 64-bit like `square_sum` (pyx:57).                             */
/* ------------------------------------------------------------------------- */
ORC_API int orc_sauvola(const uint8_t *in, uint8_t *out, int w, int h,
                        int ww, int wh, double k, double R)
{
    const double k2 = k * k / R / R;
    const double km1 = k - 1;
    const int l = (ww + 1) / 2, r = ww / 2, o = (wh + 1) / 2, u = wh / 2;
    int32_t *cs = (int32_t *)calloc((size_t)w, sizeof(int32_t));
    int32_t *cq = (int32_t *)calloc((size_t)w, sizeof(int32_t));
    if (!cs || !cq) { free(cs); free(cq); return -1; }

    /* rows [0, min(h,u)) are in the window of the (virtual) row -1 */
    int rows_in = imin(h, u);
    for (int y = 0; y < rows_in; y++)
        for (int x = 0; x < w; x++) {
            int32_t p = in[(size_t)y * w + x];
            cs[x] += p; cq[x] += p * p;
        }
    for (int y = 0; y < h; y++) {
        int leave = y - o;         /* row leaving the window */
        int enter = y + u;         /* row entering */
        if (leave >= 0)
            for (int x = 0; x < w; x++) {
                int32_t p = in[(size_t)leave * w + x];
                cs[x] -= p; cq[x] -= p * p;
            }
        if (enter < h)
            for (int x = 0; x < w; x++) {
                int32_t p = in[(size_t)enter * w + x];
                cs[x] += p; cq[x] += p * p;
            }
        int top = leave >= 0 ? leave : -1;
        int bot = enter < h ? enter : h - 1;
        int nrows = bot - top;
        /* horizontal sliding window over column sums */
        int32_t S = 0; int64_t Q = 0;
        int hi = -1;               /* columns [lo, hi] currently summed */
        int lo = 0;
        const uint8_t *row = in + (size_t)y * w;
        uint8_t *orow = out + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int want_hi = imin(x + r, w - 1);
            int want_lo = imax(x - l + 1, 0);
            while (hi < want_hi) { hi++; S += cs[hi]; Q += cq[hi]; }
            while (lo < want_lo) { S -= cs[lo]; Q -= cq[lo]; lo++; }
            int count = (want_hi - want_lo + 1) * nrows;
            double mean = (double)(S / count);
            double variance = (double)(Q / count) - mean * mean;
            double tmp = (double)row[x] + mean * km1;
            int form;
            if (k >= 0)
                form = (tmp <= 0) || (tmp * tmp <= mean * mean * k2 * variance);
            else
                form = (tmp <= 0) && (tmp * tmp >= mean * mean * k2 * variance);
            orow[x] = form ? 0 : 1;
        }
    }
    free(cs); free(cq);
    return 0;
}

/* a2: mrc.threshold_image (mrc.py:58-87): square window, R=128, result inverted
 * so that 1 = dark = foreground.  `img` may be a strided crop (the reference
 * reshapes, which copies, mrc.py:80). */
ORC_API int orc_window_for_dpi(int has_dpi, double dpi)
{
    int window = 51;                           /* mrc.py:68 */
    if (has_dpi) {
        window = (int)(dpi / 4);               /* mrc.py:71 */
        if (window % 2 == 0) window += 1;      /* mrc.py:72-73 */
    }
    return window;
}

ORC_API int orc_threshold_image(const uint8_t *img, int stride, int w, int h,
                                int window, double k, int invert_input,
                                uint8_t *out)
{
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h + 1);
    if (!tmp) return -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t p = img[(size_t)y * stride + x];
            tmp[(size_t)y * w + x] = invert_input ? (uint8_t)(255 - p) : p; /* mrc.py:224 */
        }
    int rc = orc_sauvola(tmp, out, w, h, window, window, k, 128.0);          /* mrc.py:82 */
    for (size_t i = 0; i < (size_t)w * h; i++) out[i] = out[i] ? 0 : 1;     /* mrc.py:85 */
    free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* a6: optimiser.fast_mask_denoise (cython/optimiser.pyx:436-472)            */
/* raster order, in place, reads already-updated neighbours.                 */
/* ------------------------------------------------------------------------- */
ORC_API void orc_denoise(uint8_t *mask, int w, int h, int mincnt, int n)
{
    for (int y = n; y < h - n; y++)
        for (int x = n; x < w - n; x++) {
            uint8_t *p = mask + (size_t)y * w + x;
            if (!*p) continue;
            int cnt = 0;
            for (int dy = -n; dy <= n; dy++)
                for (int dx = -n; dx <= n; dx++)
                    cnt += p[(ptrdiff_t)dy * w + dx];
            *p = (uint8_t)((cnt - 1) >= mincnt);     /* pyx:470 */
        }
}

/* ------------------------------------------------------------------------- */
/* a7: optimiser.optimise_gray2 / optimise_rgb2 (optimiser.pyx:153-273,      */
/* 280-429), spec = optimise_gray / optimise_rgb (pyx:22-76, 83-146).        */
/* For every pixel with mask==0 (raster order):                              */
/*   ys=max(0,y-n) ye=min(H,y+n) xs=max(0,x-n) xe=min(W,x+n)   (half open)   */
/*   val = sum_{[ys,ye)x[xs,xe), mask!=0} img + sum_{[ys,y)x[xs,x)} out      */
/*   cnt = #mask in window + (y-ys)*(x-xs)                                   */
/*   out = cnt>0 ? val/cnt : 0                                               */
/* Restated with per-row prefix sums over per-column running sums.           */
/* invert_mask=1 treats mask as (mask ^ 1) (mrc.py:439).                     */
/* ------------------------------------------------------------------------- */
ORC_API int orc_optimise(const uint8_t *mask, const uint8_t *img, uint8_t *out,
                         int w, int h, int c, int n, int invert_mask)
{
    if (c != 1 && c != 3) return -2;
    size_t W1 = (size_t)w + 1;
    int32_t *fir = (int32_t *)calloc((size_t)w * c, sizeof(int32_t));
    int32_t *fcnt = (int32_t *)calloc((size_t)w, sizeof(int32_t));
    int32_t *iir = (int32_t *)calloc((size_t)w * c, sizeof(int32_t));
    int32_t *pf = (int32_t *)malloc(W1 * c * sizeof(int32_t));
    int32_t *pm = (int32_t *)malloc(W1 * sizeof(int32_t));
    int32_t *pi = (int32_t *)malloc(W1 * c * sizeof(int32_t));
    if (!fir || !fcnt || !iir || !pf || !pm || !pi) {
        free(fir); free(fcnt); free(iir); free(pf); free(pm); free(pi);
        return -1;
    }
    memcpy(out, img, (size_t)w * h * c);          /* new_img = np.copy(img) */
    int f_lo = 0, f_hi = 0;   /* FIR rows [f_lo, f_hi) accumulated */
    int i_lo = 0, i_hi = 0;   /* IIR rows [i_lo, i_hi) accumulated */
#define MASKED(yy, xx) ((mask[(size_t)(yy) * w + (xx)] != 0) != (invert_mask != 0))
    for (int y = 0; y < h; y++) {
        int ys = imax(0, y - n), ye = imin(h, y + n);
        for (; f_lo < ys; f_lo++)
            for (int x = 0; x < w; x++)
                if (MASKED(f_lo, x)) {
                    for (int ch = 0; ch < c; ch++)
                        fir[(size_t)x * c + ch] -= img[((size_t)f_lo * w + x) * c + ch];
                    fcnt[x]--;
                }
        for (; f_hi < ye; f_hi++)
            for (int x = 0; x < w; x++)
                if (MASKED(f_hi, x)) {
                    for (int ch = 0; ch < c; ch++)
                        fir[(size_t)x * c + ch] += img[((size_t)f_hi * w + x) * c + ch];
                    fcnt[x]++;
                }
        for (; i_lo < ys; i_lo++)
            for (size_t j = 0; j < (size_t)w * c; j++)
                iir[j] -= out[(size_t)i_lo * w * c + j];
        for (; i_hi < y; i_hi++)
            for (size_t j = 0; j < (size_t)w * c; j++)
                iir[j] += out[(size_t)i_hi * w * c + j];
        /* prefix sums along the row */
        pm[0] = 0;
        for (int ch = 0; ch < c; ch++) { pf[ch] = 0; pi[ch] = 0; }
        for (int x = 0; x < w; x++) {
            pm[x + 1] = pm[x] + fcnt[x];
            for (int ch = 0; ch < c; ch++) {
                pf[(size_t)(x + 1) * c + ch] = pf[(size_t)x * c + ch] + fir[(size_t)x * c + ch];
                pi[(size_t)(x + 1) * c + ch] = pi[(size_t)x * c + ch] + iir[(size_t)x * c + ch];
            }
        }
        for (int x = 0; x < w; x++) {
            if (MASKED(y, x)) continue;
            int xs = imax(0, x - n), xe = imin(w, x + n);
            int cnt = pm[xe] - pm[xs] + (y - ys) * (x - xs);
            for (int ch = 0; ch < c; ch++) {
                int val = pf[(size_t)xe * c + ch] - pf[(size_t)xs * c + ch]
                        + pi[(size_t)x * c + ch] - pi[(size_t)xs * c + ch];
                out[((size_t)y * w + x) * c + ch] = cnt > 0 ? (uint8_t)(val / cnt) : 0;
            }
        }
    }
#undef MASKED
    free(fir); free(fcnt); free(iir); free(pf); free(pm); free(pi);
    return 0;
}

/* Spec version (optimiser.pyx:22-76 / 83-146), O(n^2) per pixel; small inputs only. */
ORC_API int orc_optimise_spec(const uint8_t *mask, const uint8_t *img, uint8_t *out,
                              int w, int h, int c, int n, int invert_mask)
{
    if (c != 1 && c != 3) return -2;
    memcpy(out, img, (size_t)w * h * c);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int m = (mask[(size_t)y * w + x] != 0) != (invert_mask != 0);
            if (m) continue;
            int ys = imax(0, y - n), ye = imin(h, y + n);
            int xs = imax(0, x - n), xe = imin(w, x + n);
            int cnt = 0, val[3] = {0, 0, 0};
            for (int yy = ys; yy < ye; yy++)
                for (int xx = xs; xx < xe; xx++)
                    if ((mask[(size_t)yy * w + xx] != 0) != (invert_mask != 0)) {
                        for (int ch = 0; ch < c; ch++) val[ch] += img[((size_t)yy * w + xx) * c + ch];
                        cnt++;
                    }
            for (int yy = ys; yy < y; yy++)
                for (int xx = xs; xx < x; xx++) {
                    for (int ch = 0; ch < c; ch++) val[ch] += out[((size_t)yy * w + xx) * c + ch];
                    cnt++;
                }
            for (int ch = 0; ch < c; ch++)
                out[((size_t)y * w + x) * c + ch] = cnt > 0 ? (uint8_t)(val[ch] / cnt) : 0;
        }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a10: PIL Image.convert('L') for RGB (mrc.py:361).  Pillow Convert.c uses   */
/* the ITU-R 601-2 luma in 16.16 fixed point with rounding:                  */
/*   L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16                         */
/* ------------------------------------------------------------------------- */
ORC_API void orc_luma601(const uint8_t *rgb, uint8_t *gray, size_t npx)
{
    for (size_t i = 0; i < npx; i++) {
        uint32_t v = rgb[3 * i] * 19595u + rgb[3 * i + 1] * 38470u + rgb[3 * i + 2] * 7471u + 0x8000u;
        gray[i] = (uint8_t)(v >> 16);
    }
}

/* ------------------------------------------------------------------------- */
/* a8: skimage.restoration.estimate_sigma -> pywt.dwtn(x,'db2')['dd']        */
/* (mrc.py:52-55).  One-axis db2 high-pass analysis with downsampling,       */
/* mode 'symmetric' (half-sample mirror), as PyWavelets' convolution:        */
/*   y[k] = sum_{j=0..3} f[j] * xsym[2k+1-j],  k = 0 .. (N+3)/2 - 1          */
/* accumulated in the array's precision starting from 0, no FMA.  For        */
/* outputs with 2k+1 >= N the mirrored (right extension) products come first */
/* (filter index descending), then the in-range products ascending.          */
/* float32 arrays use the float32-rounded filter.                            */
/* ------------------------------------------------------------------------- */
static const double DB2_HI[4] = {
    -0.48296291314453416, 0.8365163037378079, -0.2241438680420134, -0.12940952255126037
};

#define DEF_DWT1D(NAME, T)                                                         \
static void NAME(const T *x, ptrdiff_t xs, int N, T *y, ptrdiff_t ys, const T *f)   \
{                                                                                  \
    const int F = 4;                                                               \
    int o = 0;                                                                     \
    for (int i = 1; i < N + F - 1; i += 2, o++) {                                  \
        T sum = 0;                                                        \
        if (i < N) {                                                               \
            int j = 0;                                                             \
            for (; j <= i && j < F; j++) sum += f[j] * x[(ptrdiff_t)(i - j) * xs]; \
            /* left extension: x[-1-t] = x[t] (periodic mirror for tiny N) */      \
            while (j < F) {                                                        \
                int kk;                                                            \
                for (kk = 0; kk < N && j < F; j++, kk++) sum += f[j] * x[(ptrdiff_t)kk * xs]; \
                for (kk = 0; kk < N && j < F; j++, kk++) sum += f[j] * x[(ptrdiff_t)(N - 1 - kk) * xs]; \
            }                                                                      \
        } else {                                                                   \
            int j = 0;                                                             \
            /* right extension first: x[N+t] = x[N-1-t] */                         \
            while (i - j >= N) {                                                   \
                int kk;                                                            \
                for (kk = 0; kk < N && i - j >= N; j++, kk++)                      \
                    sum += f[i - N - j] * x[(ptrdiff_t)(N - 1 - kk) * xs];         \
                for (kk = 0; kk < N && i - j >= N; j++, kk++)                      \
                    sum += f[i - N - j] * x[(ptrdiff_t)kk * xs];                   \
            }                                                                      \
            /* here j = i-N+1; remaining taps j..F-1 index x[i-j] in range */      \
            for (; j <= i && j < F; j++) sum += f[j] * x[(ptrdiff_t)(i - j) * xs]; \
            while (j < F) {                                                        \
                int kk;                                                            \
                for (kk = 0; kk < N && j < F; j++, kk++) sum += f[j] * x[(ptrdiff_t)kk * xs]; \
                for (kk = 0; kk < N && j < F; j++, kk++) sum += f[j] * x[(ptrdiff_t)(N - 1 - kk) * xs]; \
            }                                                                      \
        }                                                                          \
        y[(ptrdiff_t)o * ys] = sum;                                                \
    }                                                                              \
}
DEF_DWT1D(dwt1d_f32, float)
DEF_DWT1D(dwt1d_f64, double)

ORC_API int orc_dwt_len(int n) { return (n + 3) / 2; }

/* dd = high-pass along axis 0 then along axis 1; dd is [(h+3)/2][(w+3)/2] */
ORC_API int orc_dwt_dd_f32(const float *x, int stride, int h, int w, float *dd)
{
    float f[4];
    for (int i = 0; i < 4; i++) f[i] = (float)DB2_HI[i];
    int h2 = orc_dwt_len(h), w2 = orc_dwt_len(w);
    float *t = (float *)malloc((size_t)h2 * w * sizeof(float));
    if (!t) return -1;
    for (int xx = 0; xx < w; xx++) dwt1d_f32(x + xx, stride, h, t + xx, w, f);
    for (int yy = 0; yy < h2; yy++) dwt1d_f32(t + (size_t)yy * w, 1, w, dd + (size_t)yy * w2, 1, f);
    free(t);
    return 0;
}

ORC_API int orc_dwt_dd_f64(const double *x, int stride, int h, int w, double *dd)
{
    int h2 = orc_dwt_len(h), w2 = orc_dwt_len(w);
    double *t = (double *)malloc((size_t)h2 * w * sizeof(double));
    if (!t) return -1;
    for (int xx = 0; xx < w; xx++) dwt1d_f64(x + xx, stride, h, t + xx, w, DB2_HI);
    for (int yy = 0; yy < h2; yy++) dwt1d_f64(t + (size_t)yy * w, 1, w, dd + (size_t)yy * w2, 1, DB2_HI);
    free(t);
    return 0;
}

static int cmp_f32(const void *a, const void *b)
{ float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
static int cmp_f64(const void *a, const void *b)
{ double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

#define SIGMA_DENOM 0.6744897501960817   /* scipy.stats.norm.ppf(0.75) */

/* skimage _sigma_est_dwt: median(|dd[dd != 0]|) / ppf(0.75); np.median of a
 * float32 array returns the float32 mean of the two middle values for even
 * counts; the division by the float64 denominator promotes to float64.
 * Empty selection -> NaN. */
ORC_API double orc_sigma_f32(const float *x, int stride, int h, int w)
{
    int h2 = orc_dwt_len(h), w2 = orc_dwt_len(w);
    size_t n = (size_t)h2 * w2, m = 0;
    float *dd = (float *)malloc(n * sizeof(float));
    if (!dd) return NAN;
    orc_dwt_dd_f32(x, stride, h, w, dd);
    for (size_t i = 0; i < n; i++) if (dd[i] != 0.0f) dd[m++] = fabsf(dd[i]);
    double res;
    if (m == 0) res = NAN;
    else {
        qsort(dd, m, sizeof(float), cmp_f32);
        float med;
        if (m & 1) med = dd[m / 2];
        else { float s = dd[m / 2 - 1] + dd[m / 2]; med = s / 2.0f; }
        res = (double)med / SIGMA_DENOM;
    }
    free(dd);
    return res;
}

/* bool arrays take PyWavelets' float64 path (mrc.py:253-254). */
ORC_API double orc_sigma_bool(const uint8_t *b, int stride, int h, int w)
{
    int h2 = orc_dwt_len(h), w2 = orc_dwt_len(w);
    size_t n = (size_t)h2 * w2, m = 0;
    double *x = (double *)malloc((size_t)h * w * sizeof(double));
    double *dd = (double *)malloc(n * sizeof(double));
    if (!x || !dd) { free(x); free(dd); return NAN; }
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++) x[(size_t)yy * w + xx] = b[(size_t)yy * stride + xx] ? 1.0 : 0.0;
    orc_dwt_dd_f64(x, w, h, w, dd);
    for (size_t i = 0; i < n; i++) if (dd[i] != 0.0) dd[m++] = fabs(dd[i]);
    double res;
    if (m == 0) res = NAN;
    else {
        qsort(dd, m, sizeof(double), cmp_f64);
        double med = (m & 1) ? dd[m / 2] : (dd[m / 2 - 1] + dd[m / 2]) / 2.0;
        res = med / SIGMA_DENOM;
    }
    free(x); free(dd);
    return res;
}

/* a4: mrc.estimate_noise (mrc.py:273-296): central half crop. */
ORC_API void orc_noise_crop(int h, int w, int *hs, int *he, int *ws, int *we)
{
    *hs = (int)(h / 2.0 - h / 4.0); *he = (int)(h / 2.0 + h / 4.0);   /* mrc.py:282-285 */
    *ws = (int)(w / 2.0 - w / 4.0); *we = (int)(w / 2.0 + w / 4.0);
    if (*he == 0 || *we == 0) { *hs = 0; *he = h; *ws = 0; *we = w; } /* mrc.py:288-292 */
}

ORC_API double orc_estimate_noise(const float *imgf, int h, int w)
{
    int hs, he, ws, we;
    orc_noise_crop(h, w, &hs, &he, &ws, &we);
    return orc_sigma_f32(imgf + (size_t)hs * w + ws, w, he - hs, we - ws);
}

/* ------------------------------------------------------------------------- */
/* a9: scipy.ndimage.gaussian_filter (mrc.py:311).                            */
/* radius = int(4*sigma + 0.5); weights exp(-0.5/sigma^2 * x^2) normalised    */
/* by their numpy sum (pairwise summation order for >= 8 elements).          */
/* correlate1d with a symmetric kernel: per output                           */
/*   acc = x[c]*w[r];  for j=-r..-1: acc += (x[c+j] + x[c-j]) * w[r+j]       */
/* in float64, borders 'reflect' (d c b a | a b c d | d c b a), axis 0 then  */
/* axis 1, float32 between the passes.                                       */
/* NOTE numpy evaluates exp() with its own SIMD routine; libm's exp may       */
/* differ by an ulp, so callers that need numpy's table pass it in.          */
/* ------------------------------------------------------------------------- */
ORC_API int orc_gaussian_radius(double sigma) { return (int)(4.0 * sigma + 0.5); }

static double np_pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    /* n <= 128 in every realistic case (radius <= 63) */
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

ORC_API int orc_gaussian_weights(double sigma, double *wts /* 2*radius+1 */)
{
    int radius = orc_gaussian_radius(sigma);
    if (radius > 60) return -1;
    double sigma2 = sigma * sigma;
    double c = -0.5 / sigma2;
    for (int i = -radius; i <= radius; i++) wts[i + radius] = exp(c * (double)(i * i));
    double s = np_pairwise_sum(wts, 2 * radius + 1);
    for (int i = 0; i < 2 * radius + 1; i++) wts[i] = wts[i] / s;
    return radius;
}

static inline int reflect_idx(int i, int n)
{
    /* scipy NI_EXTEND_REFLECT: ... c b a | a b c ... | c b a ... (period 2n) */
    if (n == 1) return 0;
    int p = 2 * n;
    i %= p; if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

static void gauss_line(const float *src, ptrdiff_t ss, int n, float *dst, ptrdiff_t ds,
                       const double *wts, int radius, double *buf)
{
    for (int i = -radius; i < n + radius; i++) buf[i + radius] = (double)src[(ptrdiff_t)reflect_idx(i, n) * ss];
    for (int i = 0; i < n; i++) {
        const double *c = buf + radius + i;
        double acc = c[0] * wts[radius];
        for (int j = -radius; j < 0; j++) acc += (c[j] + c[-j]) * wts[radius + j];
        dst[(ptrdiff_t)i * ds] = (float)acc;
    }
}

ORC_API int orc_gaussian_f32(const float *in, float *out, int h, int w,
                             const double *wts, int radius)
{
    if (radius == 0) { memcpy(out, in, (size_t)h * w * sizeof(float)); return 0; }
    float *t = (float *)malloc((size_t)h * w * sizeof(float));
    double *buf = (double *)malloc(((size_t)imax(h, w) + 2 * radius) * sizeof(double));
    if (!t || !buf) { free(t); free(buf); return -1; }
    for (int x = 0; x < w; x++) gauss_line(in + x, w, h, t + x, w, wts, radius, buf);
    for (int y = 0; y < h; y++) gauss_line(t + (size_t)y * w, 1, w, out + (size_t)y * w, 1, wts, radius, buf);
    free(t); free(buf);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a11: PIL.Image.thumbnail((int(w/f), int(h/f)))  (mrc.py:422-428,456-462)   */
/* Pillow Image.thumbnail -> round_aspect size, reducing_gap=2.0 ->          */
/* Image.reduce (Reduce.c box mean) -> Image.resize BICUBIC (Resample.c,     */
/* 8bpc fixed point, PRECISION_BITS = 22, horizontal then vertical pass).    */
/* ------------------------------------------------------------------------- */
static int round_aspect_w(double number, double aspect, int y)
{
    /* max(min(floor, ceil, key=|aspect - n/y|), 1); min() keeps the first on ties */
    double fl = floor(number), ce = ceil(number);
    double kf = fabs(aspect - fl / y), kc = fabs(aspect - ce / y);
    double pick = (kc < kf) ? ce : fl;
    return pick < 1 ? 1 : (int)pick;
}
static int round_aspect_h(double number, double aspect, int x)
{
    double fl = floor(number), ce = ceil(number);
    double kf = fl == 0 ? 0 : fabs(aspect - x / fl), kc = ce == 0 ? 0 : fabs(aspect - x / ce);
    double pick = (kc < kf) ? ce : fl;
    return pick < 1 ? 1 : (int)pick;
}

/* returns 0 if the image is left untouched (already small enough) */
ORC_API int orc_thumbnail_size(int w, int h, int req_w, int req_h, int *ow, int *oh)
{
    int x = req_w, y = req_h;
    if (x >= w && y >= h) { *ow = w; *oh = h; return 0; }
    double aspect = (double)w / (double)h;
    if ((double)x / (double)y >= aspect) x = round_aspect_w(y * aspect, aspect, y);
    else y = round_aspect_h(x / aspect, aspect, x);
    *ow = x; *oh = y;
    return (x != w || y != h);
}

/* Reduce.c: box mean with rounding, ((ss + amend) * multiplier) >> 24,
 * multiplier = (UINT32)(2^32 / (256 * cells)) evaluated in float32,
 * amend = cells/2; partial edge cells use their actual cell count. */
static uint32_t reduce_multiplier(int cells)
{
    uint32_t max_dividend = 256u * (uint32_t)cells;
    float max_int = (float)(1 << 30) * 4.0f;
    return (uint32_t)(max_int / (float)max_dividend);
}

ORC_API void orc_reduce(const uint8_t *in, int w, int h, int c, int fx, int fy, uint8_t *out)
{
    int ow = (w + fx - 1) / fx, oh = (h + fy - 1) / fy;
    for (int oy = 0; oy < oh; oy++) {
        int y0 = oy * fy, y1 = imin(h, y0 + fy);
        for (int ox = 0; ox < ow; ox++) {
            int x0 = ox * fx, x1 = imin(w, x0 + fx);
            int cells = (y1 - y0) * (x1 - x0);
            uint32_t mult = reduce_multiplier(cells), amend = (uint32_t)cells / 2;
            for (int ch = 0; ch < c; ch++) {
                uint32_t ss = 0;
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++) ss += in[((size_t)yy * w + xx) * c + ch];
                out[((size_t)oy * ow + ox) * c + ch] = (uint8_t)(((ss + amend) * mult) >> 24);
            }
        }
    }
}

static double bicubic_filter(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

/* Resample.c sinc_filter / lanczos_filter (support 3.0): the filter of the page-ingest downsample,
 * recode.py:368-372 (Image.thumbnail(..., resample=Image.LANCZOS, reducing_gap=None)). */
static double sinc_filter(double x)
{
    if (x == 0.0) return 1.0;
    x = x * 3.14159265358979323846;     /* M_PI */
    return sin(x) / x;
}
static double lanczos_filter(double x)
{
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}
/* filter ids shared with include/mrchip.h: 0 = BICUBIC (support 2), 1 = LANCZOS (support 3) */
static double filter_support(int filter) { return filter == 1 ? 3.0 : 2.0; }
static double filter_eval(int filter, double x) { return filter == 1 ? lanczos_filter(x) : bicubic_filter(x); }

/* Resample.c precompute_coeffs + normalize_coeffs_8bpc.  in0/in1 are float32
 * (the box is passed to C as float). Returns ksize; bounds[2*xx]=xmin,
 * bounds[2*xx+1]=count; kk[xx*ksize + i] fixed-point weights. */
ORC_API int orc_resample_ksize(int filter, int in_size, float in0, float in1, int out_size)
{
    (void)in_size;
    double scale = (double)(in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = filter_support(filter) * filterscale;
    return (int)ceil(support) * 2 + 1;
}
ORC_API int orc_bicubic_ksize(int in_size, float in0, float in1, int out_size)
{
    return orc_resample_ksize(0, in_size, in0, in1, out_size);
}

ORC_API int orc_resample_coeffs(int filter, int in_size, float in0, float in1, int out_size,
                                int32_t *bounds, int32_t *kk)
{
    double scale = (double)(in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = filter_support(filter) * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    double *k = (double *)malloc((size_t)ksize * sizeof(double));
    if (!k) return -1;
    for (int xx = 0; xx < out_size; xx++) {
        double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; x++) {
            double wv = filter_eval(filter, (x + xmin - center + 0.5) * ss);
            k[x] = wv; ww += wv;
        }
        for (x = 0; x < xmax; x++) if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; x++) k[x] = 0;
        for (x = 0; x < ksize; x++) {
            double v = k[x] * (double)(1 << 22);
            kk[(size_t)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v) : (int32_t)(0.5 + v);
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    free(k);
    return ksize;
}

ORC_API int orc_bicubic_coeffs(int in_size, float in0, float in1, int out_size, int32_t *bounds, int32_t *kk)
{
    return orc_resample_coeffs(0, in_size, in0, in1, out_size, bounds, kk);
}

static inline uint8_t clip8(int32_t v)
{
    v >>= 22;                       /* arithmetic shift, as Pillow's clip8 lookup */
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}

/* Image.resize(size, filter, box=(0,0,bw,bh)) on 8-bit images. */
ORC_API int orc_resize(int filter, const uint8_t *in, int w, int h, int c,
                       float box_w, float box_h, int ow, int oh, uint8_t *out)
{
    int need_h = (ow != w) || (box_w != (float)ow);
    int need_v = (oh != h) || (box_h != (float)oh);
    const uint8_t *src = in;
    uint8_t *tmp = NULL;
    int cur_w = w;
    if (need_h) {
        int ks = orc_resample_ksize(filter, w, 0.f, box_w, ow);
        int32_t *b = (int32_t *)malloc((size_t)ow * 2 * sizeof(int32_t));
        int32_t *kk = (int32_t *)malloc((size_t)ow * ks * sizeof(int32_t));
        tmp = (uint8_t *)malloc((size_t)ow * h * c);
        if (!b || !kk || !tmp) { free(b); free(kk); free(tmp); return -1; }
        orc_resample_coeffs(filter, w, 0.f, box_w, ow, b, kk);
        for (int y = 0; y < h; y++)
            for (int xx = 0; xx < ow; xx++) {
                int xmin = b[2 * xx], n = b[2 * xx + 1];
                const int32_t *kx = kk + (size_t)xx * ks;
                for (int ch = 0; ch < c; ch++) {
                    int32_t ss = 1 << 21;
                    for (int x = 0; x < n; x++) ss += in[((size_t)y * w + xmin + x) * c + ch] * kx[x];
                    tmp[((size_t)y * ow + xx) * c + ch] = clip8(ss);
                }
            }
        free(b); free(kk);
        src = tmp; cur_w = ow;
    }
    if (need_v) {
        int ks = orc_resample_ksize(filter, h, 0.f, box_h, oh);
        int32_t *b = (int32_t *)malloc((size_t)oh * 2 * sizeof(int32_t));
        int32_t *kk = (int32_t *)malloc((size_t)oh * ks * sizeof(int32_t));
        if (!b || !kk) { free(b); free(kk); free(tmp); return -1; }
        orc_resample_coeffs(filter, h, 0.f, box_h, oh, b, kk);
        for (int yy = 0; yy < oh; yy++) {
            int ymin = b[2 * yy], n = b[2 * yy + 1];
            const int32_t *ky = kk + (size_t)yy * ks;
            for (size_t j = 0; j < (size_t)cur_w * c; j++) {
                int32_t ss = 1 << 21;
                for (int y = 0; y < n; y++) ss += src[((size_t)(ymin + y) * cur_w) * c + j] * ky[y];
                out[(size_t)yy * cur_w * c + j] = clip8(ss);
            }
        }
        free(b); free(kk);
    } else {
        memcpy(out, src, (size_t)cur_w * h * c);
    }
    free(tmp);
    return 0;
}

ORC_API int orc_resize_bicubic(const uint8_t *in, int w, int h, int c,
                               float box_w, float box_h, int ow, int oh, uint8_t *out)
{
    return orc_resize(0, in, w, h, c, box_w, box_h, ow, oh, out);
}

/* Image.thumbnail((req_w, req_h), resample=filter, reducing_gap=gap); gap <= 0 means None (no
 * Image.reduce step).  out must hold ow*oh*c bytes with (ow,oh) from orc_thumbnail_size. */
ORC_API int orc_thumbnail_ex(const uint8_t *in, int w, int h, int c, int req_w, int req_h, int filter,
                             double gap, uint8_t *out)
{
    int ow, oh;
    if (!orc_thumbnail_size(w, h, req_w, req_h, &ow, &oh)) {
        memcpy(out, in, (size_t)w * h * c);
        return 0;
    }
    /* Image.resize: factor = int(box_extent / size / reducing_gap) or 1 */
    int fx = 1, fy = 1;
    if (gap > 0) {
        fx = (int)((double)w / ow / gap); if (fx < 1) fx = 1;
        fy = (int)((double)h / oh / gap); if (fy < 1) fy = 1;
    }
    if (fx > 1 || fy > 1) {
        int rw = (w + fx - 1) / fx, rh = (h + fy - 1) / fy;
        uint8_t *red = (uint8_t *)malloc((size_t)rw * rh * c);
        if (!red) return -1;
        orc_reduce(in, w, h, c, fx, fy, red);
        float bw = (float)((double)w / fx), bh = (float)((double)h / fy);
        int rc = orc_resize(filter, red, rw, rh, c, bw, bh, ow, oh, out);
        free(red);
        return rc;
    }
    return orc_resize(filter, in, w, h, c, (float)w, (float)h, ow, oh, out);
}

/* Image.thumbnail((req_w, req_h)) with the defaults BICUBIC / reducing_gap=2.0 (mrc.py:422-428). */
ORC_API int orc_thumbnail(const uint8_t *in, int w, int h, int c, int req_w, int req_h, uint8_t *out)
{
    return orc_thumbnail_ex(in, w, h, c, req_w, req_h, 0, 2.0, out);
}

/* ------------------------------------------------------------------------- */
/* a3: mrc.create_hocr_mask (mrc.py:188-270), boxes already validated and     */
/* converted to int by the host (mrc.py:198-221 is text/confidence logic).   */
/* boxes = [l,t,r,b]*nb in list order.  decisions (optional, nb ints) get    */
/* 0 = none, 1 = thres, 2 = thres_invert.                                    */
/* ------------------------------------------------------------------------- */
ORC_API int orc_hocr_mask(const uint8_t *gray, uint8_t *mask, int w, int h,
                          const int32_t *boxes, int nb, int window, int32_t *decisions)
{
    (void)h;
    for (int bi = 0; bi < nb; bi++) {
        int l = boxes[4 * bi], t = boxes[4 * bi + 1], r = boxes[4 * bi + 2], b = boxes[4 * bi + 3];
        int bw = r - l, bh = b - t;
        size_t size = (size_t)bw * bh;
        uint8_t *th = (uint8_t *)malloc(size), *thi = (uint8_t *)malloc(size);
        if (!th || !thi) { free(th); free(thi); return -1; }
        const uint8_t *crop = gray + (size_t)t * w + l;
        orc_threshold_image(crop, w, bw, bh, window, 0.1, 0, th);       /* mrc.py:229-230 */
        orc_threshold_image(crop, w, bw, bh, window, 0.1, 1, thi);      /* mrc.py:235 */
        size_t ones = 0, ones_i = 0;
        for (size_t i = 0; i < size; i++) { ones += th[i]; ones_i += thi[i]; }
        double ratio = (double)ones / (double)size;                      /* mrc.py:233 */
        double inv_ratio = (double)ones_i / (double)size;                /* mrc.py:238 */
        const uint8_t *pick = NULL;
        int dec = 0;
        if (ratio < 0.3 || inv_ratio < 0.3) {                            /* mrc.py:240 */
            if (inv_ratio > 0.2 && ratio < 0.2) { pick = th; dec = 1; } /* mrc.py:247-248 */
            else {
                double rs = orc_sigma_bool(th, bw, bh, bw);              /* mrc.py:253 */
                double irs = orc_sigma_bool(thi, bw, bh, bw);            /* mrc.py:254 */
                if (inv_ratio < 0.3 && inv_ratio < ratio &&
                    (irs < rs || (rs < 0.1 && irs < 0.1))) { pick = thi; dec = 2; } /* :258-261 */
                else if (ratio < 0.2) { pick = th; dec = 1; }            /* mrc.py:262-263 */
            }
        }
        if (pick)
            for (int y = 0; y < bh; y++)
                memcpy(mask + (size_t)(t + y) * w + l, pick + (size_t)y * bw, (size_t)bw); /* :266 */
        if (decisions) decisions[bi] = dec;
        free(th); free(thi);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a5 + a12: create_threshold_mask (mrc.py:300-329) and the three-yield       */
/* orchestration create_mrc_hocr_components (mrc.py:334-471) for 'L'/'RGB'.  */
/* gauss_wts: optional caller-supplied table (numpy's) for the sigma the     */
/* function will compute; pass NULL to use libm exp.                         */
/* ------------------------------------------------------------------------- */
ORC_API int orc_threshold_mask(uint8_t *mask, const uint8_t *gray, int w, int h, int window,
                               double *sigma_out, const double *gauss_wts)
{
    size_t P = (size_t)w * h;
    float *imgf = (float *)malloc(P * sizeof(float));
    uint8_t *g8 = (uint8_t *)malloc(P), *thr = (uint8_t *)malloc(P);
    if (!imgf || !g8 || !thr) { free(imgf); free(g8); free(thr); return -1; }
    for (size_t i = 0; i < P; i++) imgf[i] = (float)gray[i];              /* mrc.py:372 */
    double sigma_est = orc_estimate_noise(imgf, h, w);                    /* mrc.py:305 */
    if (sigma_out) *sigma_out = sigma_est;
    if (sigma_est > 1.0) {                                                /* mrc.py:309 */
        double sigma = sigma_est * 0.1;
        int radius = orc_gaussian_radius(sigma);
        double wts[128];
        if (radius > 60) { free(imgf); free(g8); free(thr); return -3; }
        if (gauss_wts) memcpy(wts, gauss_wts, (size_t)(2 * radius + 1) * sizeof(double));
        else orc_gaussian_weights(sigma, wts);
        float *bl = (float *)malloc(P * sizeof(float));
        if (!bl) { free(imgf); free(g8); free(thr); return -1; }
        orc_gaussian_f32(imgf, bl, h, w, wts, radius);                    /* mrc.py:311 */
        free(imgf); imgf = bl;
    }
    for (size_t i = 0; i < P; i++) g8[i] = (uint8_t)imgf[i];              /* astype(uint8), mrc.py:325 */
    orc_threshold_image(g8, w, w, h, window, 0.34, 0, thr);               /* mrc.py:325 */
    for (size_t i = 0; i < P; i++) mask[i] |= thr[i];                     /* mrc.py:329 */
    free(imgf); free(g8); free(thr);
    return 0;
}

ORC_API int orc_page_mask(const uint8_t *img, int w, int h, int c,
                          const int32_t *boxes, int nb, int window, int denoise_fast,
                          uint8_t *mask, double *sigma_out, const double *gauss_wts)
{
    size_t P = (size_t)w * h;
    uint8_t *gray = NULL;
    const uint8_t *g = img;
    if (c == 3) {
        gray = (uint8_t *)malloc(P);
        if (!gray) return -1;
        orc_luma601(img, gray, P);                                        /* mrc.py:361 */
        g = gray;
    }
    memset(mask, 0, P);                                                   /* mrc.py:367 */
    int rc = orc_hocr_mask(g, mask, w, h, boxes, nb, window, NULL);       /* mrc.py:370 */
    if (!rc) rc = orc_threshold_mask(mask, g, w, h, window, sigma_out, gauss_wts); /* mrc.py:380 */
    if (!rc && denoise_fast) orc_denoise(mask, w, h, 4, 2);               /* mrc.py:388 */
    free(gray);
    return rc;
}

/* fg (n=3, mask) / bg (n=10, inverted mask) layer incl. optional thumbnail.
 * ds <= 0 means no downsample.  Returns 1 if 'too-small-to-downsample'
 * (mrc.py:429-431 / 463-465), <0 on error.  out holds w*h*c bytes. */
ORC_API int orc_page_layer(const uint8_t *img, const uint8_t *mask, int w, int h, int c,
                           int is_bg, double ds, uint8_t *out, int *ow, int *oh)
{
    int rc = orc_optimise(mask, img, out, w, h, c, is_bg ? 10 : 3, is_bg ? 1 : 0); /* :412-415, 446-449 */
    *ow = w; *oh = h;
    if (rc) return rc;
    if (ds > 0) {
        int wd = (int)(w / ds), hd = (int)(h / ds);                       /* mrc.py:423-424 */
        if (wd > 0 && hd > 0) {
            int tw, th;
            if (orc_thumbnail_size(w, h, wd, hd, &tw, &th)) {
                uint8_t *t = (uint8_t *)malloc((size_t)tw * th * c);
                if (!t) return -1;
                rc = orc_thumbnail(out, w, h, c, wd, hd, t);
                if (!rc) { memcpy(out, t, (size_t)tw * th * c); *ow = tw; *oh = th; }
                free(t);
                return rc;
            }
        } else return 1;
    }
    return 0;
}
