/*
 * dfx_fused.h -- C ABI of the fused elementwise epilogues around the convolutions / GEMMs that
 * feed the deformable-attention path.
 *
 * The reference runs these as separate PyTorch kernels:
 *   FrozenBatchNorm2d.forward   x * scale + bias        /root/reference/models/backbone_scratch.py:58-68
 *   Bottleneck tail             out += identity; relu   (torchvision resnet50, used at backbone_scratch.py:156-159)
 * With the frozen statistics folded into the convolution weights (scale) the remaining work per
 * convolution is one pass:  y = act(x + bias[c] (+ residual)).
 *
 * Same conventions as dfx_msda.h (device pointers, caller-owned buffers, enqueue-only, 0 / <0).
 */
#ifndef DFX_FUSED_H
#define DFX_FUSED_H

#ifdef __cplusplus
extern "C" {
#endif

/* x, residual (may be NULL), out: [N,C,HW] packed (NCHW); bias [C]; out may alias x.
 * out = relu?( x + bias[c] + residual ) */
int dfx_bias_act_nchw_f32(const float *x, const float *bias, const float *residual, float *out,
                          int N, int C, long HW, int relu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_FUSED_H */
