/*
 * dfx_preprocess.h -- C ABI of the fused input preprocessing.
 *
 * One kernel per image does what the reference does on the CPU with PIL + torchvision before the
 * path (/root/reference/inference.py:285-350 ResizeWithMax/resize, :422-450 transform stacks,
 * util/misc.py:338-356 padding collate):
 *   bilinear resize with Pillow's resampler semantics (22-bit fixed-point taps, horizontal pass then
 *   vertical pass, uint8 intermediate) -> /255 -> (x - mean[c]) / std[c] -> written into the frame's
 *   slot of the zero-padded batch tensor, plus the padding mask (1 = padding).
 * The taps are computed on the host from the two sizes (models/preprocess.py, same formulas as
 * Pillow's precompute_coeffs / normalize_coeffs_8bpc) and handed over as device arrays:
 *   xbounds int32 [Wo,2] = (first source column, tap count), xcoef int32 [Wo,kx]; kx = 0 means
 *   "no horizontal pass" (Wo == Ws); likewise ybounds / ycoef / ky for rows.
 *   src  uint8 [Hs,Ws,Cs] (HWC, Cs <= 4)          mean, std float32 [Cs]
 *   dst  float32, channel planes of Hp x Wp: channel c of this image goes to dst + c * plane_stride
 *   mask uint8 [Hp,Wp] or NULL
 * Same conventions as dfx_msda.h (device pointers, enqueue-only, 0 / <0).
 */
#ifndef DFX_PREPROCESS_H
#define DFX_PREPROCESS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int dfx_preprocess_u8_f32(const uint8_t *src, int Hs, int Ws, int Cs,
                          const int32_t *xbounds, const int32_t *xcoef, int kx,
                          const int32_t *ybounds, const int32_t *ycoef, int ky,
                          int Ho, int Wo, const float *mean, const float *std,
                          float *dst, long plane_stride, int Hp, int Wp, uint8_t *mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_PREPROCESS_H */
