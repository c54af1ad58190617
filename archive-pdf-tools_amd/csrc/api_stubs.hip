// Entry points whose kernels are not wired yet (temporary; shrinks as stages land).
#include "mrchip_internal.h"
using namespace mrchip;
#define STUB(name) do { set_error(name ": not implemented yet"); return MRCHIP_E_UNSUPPORTED; } while (0)

MRCHIP_EXPORT int mrchip_mask_denoise(mrchip_ctx *, uint8_t *, int, int, int, int) { STUB("mrchip_mask_denoise"); }
MRCHIP_EXPORT int mrchip_optimise(mrchip_ctx *, const uint8_t *, const uint8_t *, uint8_t *, int, int, int, int, int) { STUB("mrchip_optimise"); }
MRCHIP_EXPORT int mrchip_estimate_sigma(mrchip_ctx *, const uint8_t *, int, int, int, int, double *) { STUB("mrchip_estimate_sigma"); }
MRCHIP_EXPORT int mrchip_estimate_noise_u8(mrchip_ctx *, const uint8_t *, int, int, double *) { STUB("mrchip_estimate_noise_u8"); }
MRCHIP_EXPORT int mrchip_gaussian_u8(mrchip_ctx *, const uint8_t *, uint8_t *, int, int, double, const double *, int) { STUB("mrchip_gaussian_u8"); }
MRCHIP_EXPORT int mrchip_thumbnail_size(int, int, int, int, int *, int *) { STUB("mrchip_thumbnail_size"); }
MRCHIP_EXPORT int mrchip_thumbnail(mrchip_ctx *, const uint8_t *, int, int, int, int, int, uint8_t *) { STUB("mrchip_thumbnail"); }
MRCHIP_EXPORT int mrchip_hocr_mask(mrchip_ctx *, const uint8_t *, uint8_t *, int, int, const int32_t *, int, int, int32_t *) { STUB("mrchip_hocr_mask"); }
MRCHIP_EXPORT mrchip_page *mrchip_page_create(mrchip_ctx *, int, int, int) { set_error("page api: not implemented yet"); return nullptr; }
MRCHIP_EXPORT void mrchip_page_destroy(mrchip_page *) {}
MRCHIP_EXPORT int mrchip_page_upload(mrchip_page *, const uint8_t *) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_mask_begin(mrchip_page *, const int32_t *, int, int) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_sigma(mrchip_page *, double *) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_mask_finish(mrchip_page *, const double *, int, int) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_download_mask(mrchip_page *, uint8_t *) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_layer(mrchip_page *, int, double, int *, int *, int *) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_download_layer(mrchip_page *, int, uint8_t *) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_sync(mrchip_page *) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_box_decisions(mrchip_page *, int32_t *, int) { STUB("page"); }
MRCHIP_EXPORT int mrchip_page_device_ptrs(mrchip_page *, void **, void **, size_t *, void **, void **) { STUB("page"); }
