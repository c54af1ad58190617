/* TEST INFRASTRUCTURE (parity oracle): a whole page through the C restatement under AddressSanitizer +
 * UndefinedBehaviorSanitizer (`make -C oracle asan`; SURVEY.md 5 "race detection / sanitizers": the CPU build only --
 * GPU sanitizers are not available on this pool).  tests/test_oracle_asan.py writes the page, runs this program and
 * compares what it returns with the page goldens made by the reference.
 *
 *   asan_driver IN OUT
 * IN : int32 w h c nb window denoise_fast ; float64 fg_ds bg_ds ; uint8 img[w*h*c] ; int32 boxes[nb*4]
 * OUT: uint8 mask[w*h] ; int32 fw fh ; uint8 fg[fw*fh*c] ; int32 bw bh ; uint8 bg[bw*bh*c]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int orc_page_mask(const uint8_t *img, int w, int h, int c, const int32_t *boxes, int nb, int window, int denoise_fast,
                  uint8_t *mask, double *sigma_out, const double *gauss_wts);
int orc_page_layer(const uint8_t *img, const uint8_t *mask, int w, int h, int c, int is_bg, double ds, uint8_t *out,
                   int *ow, int *oh);

static void need(int ok, const char *what) {
    if (!ok) { fprintf(stderr, "asan_driver: %s\n", what); exit(2); }
}

int main(int argc, char **argv) {
    need(argc == 3, "usage: asan_driver IN OUT");
    FILE *f = fopen(argv[1], "rb");
    need(f != NULL, "cannot open IN");
    int32_t hd[6];
    double ds[2];
    need(fread(hd, sizeof hd, 1, f) == 1 && fread(ds, sizeof ds, 1, f) == 1, "short header");
    const int w = hd[0], h = hd[1], c = hd[2], nb = hd[3], window = hd[4], dn = hd[5];
    need(w > 0 && h > 0 && (c == 1 || c == 3) && nb >= 0, "bad header");
    const size_t P = (size_t)w * h;
    uint8_t *img = malloc(P * c), *mask = malloc(P), *out = malloc(P * c);
    int32_t *boxes = malloc(sizeof(int32_t) * 4 * (nb ? nb : 1));
    need(img && mask && out && boxes, "out of memory");
    need(fread(img, 1, P * c, f) == P * c, "short image");
    need(nb == 0 || fread(boxes, sizeof(int32_t) * 4, nb, f) == (size_t)nb, "short boxes");
    fclose(f);
    double sigma = 0;
    need(orc_page_mask(img, w, h, c, boxes, nb, window, dn, mask, &sigma, NULL) == 0, "orc_page_mask failed");
    f = fopen(argv[2], "wb");
    need(f != NULL, "cannot open OUT");
    fwrite(mask, 1, P, f);
    for (int is_bg = 0; is_bg < 2; is_bg++) {
        int ow = 0, oh = 0;
        const int rc = orc_page_layer(img, mask, w, h, c, is_bg, ds[is_bg], out, &ow, &oh);
        need(rc >= 0, "orc_page_layer failed");
        int32_t sz[2] = {ow, oh};
        fwrite(sz, sizeof sz, 1, f);
        fwrite(out, 1, (size_t)ow * oh * c, f);
    }
    fclose(f);
    free(img); free(mask); free(out); free(boxes);
    return 0;
}
