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

/* ResNet stem epilogue in one pass: out = maxpool3x3/s2/p1( relu(x + bias[c]) ), NCHW.
 * Replaces FrozenBatchNorm2d (shift; the scale is folded into conv1) + ReLU + nn.MaxPool2d of the stem
 * (/root/reference/models/backbone_scratch.py:112-115).  x [N,C,H,W] -> out [N,C,(H+1)/2,(W+1)/2]. */
int dfx_bias_relu_maxpool_f32(const float *x, const float *bias, float *out, int N, int C, int H, int W,
                              void *stream);

/* Residual add + LayerNorm over the last dimension in one pass (the reference runs ``norm(x + y)`` as an
 * add kernel followed by nn.LayerNorm: deformable_transformer_single.py:556-562 and every other block):
 *   out[r,:] = LayerNorm(x[r,:] (+ res[r,:])) * gamma + beta,  biased variance, eps inside the sqrt.
 * x, res (may be NULL), out: [rows, C] packed fp32; C a multiple of 4, C <= 1024; out may alias x. */
int dfx_add_layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta, float *out,
                          long rows, int C, float eps, void *stream);

/* Box refinement of the iterative decoders and detection heads in one pass:
 *   out[r,c] = sigmoid(delta[r,c] + inverse_sigmoid(ref[r,c]))  for c < ref_dim,  sigmoid(delta[r,c]) otherwise,
 * inverse_sigmoid(x) = log(max(clamp(x,0,1), eps) / max(1 - clamp(x,0,1), eps))
 * (/root/reference/util/misc.py inverse_sigmoid; deformable_transformer_single.py:724-735,
 * deformable_detr_single.py:203-212).  delta, out [rows,4]; ref [rows,ref_dim], ref_dim 2 or 4. */
int dfx_box_refine_f32(const float *delta, const float *ref, int ref_dim, float *out, long rows, float eps,
                       void *stream);

/* nn.GroupNorm(groups, C) of the detector's input projections (Conv1x1 + GroupNorm(32, 256),
 * /root/reference/models/deformable_detr_single.py:101-125,143-150): biased variance over each group's C/groups
 * channels x HW pixels, eps inside the sqrt, per-channel affine.
 *   x [N,C,HW] NCHW;  stats [N*groups*2] scratch (mean, rstd), caller-owned
 *   y: tokens_out = 0 -> [N,C,HW];  tokens_out = 1 -> [N,HW,C]  (the layout the transformer flattens to, so the
 *      reference's separate flatten(2).transpose(1,2) copy is not needed) */
int dfx_group_norm_f32(const float *x, const float *gamma, const float *beta, float *stats, float *y,
                       int N, int C, long HW, int groups, float eps, int tokens_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_FUSED_H */
