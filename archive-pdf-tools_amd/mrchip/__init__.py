"""mrchip -- MI355X-native MRC page decomposition (drop-in for the hot path of
internetarchive/archive-pdf-tools: internetarchivepdf/mrc.py + cython/sauvola.pyx
+ cython/optimiser.pyx).

Submodules mirror the reference's module names:
  mrchip.sauvola    binarise_sauvola                       (cython/sauvola.pyx)
  mrchip.optimiser  optimise_gray2/rgb2/gray/rgb, fast_mask_denoise (cython/optimiser.pyx)
  mrchip.mrc        threshold_image, create_hocr_mask, estimate_noise, mean_estimate_sigma,
                    create_threshold_mask, denoise_bregman (host passthrough to scikit-image),
                    create_mrc_hocr_components (internetarchivepdf/mrc.py); plus the batch forms
                    decompose_pages / decompose_stream over many pages at once

All pixel work runs in hand-written HIP kernels behind the C ABI of
include/mrchip.h (libmrchip.so, loaded with ctypes).  There is no CPU fallback:
importing works anywhere, but every compute call raises if the library or a
GPU is missing.
"""
__version__ = '0.1.0'
