/*
 * dfx_conv.h -- C ABI of the hand-written fp32 convolutions (gfx950, v_mfma_f32_32x32x2_f32) of the
 * backbones that feed the deformable-attention path.
 *
 * They stand where the reference calls cuDNN through torch.nn.Conv2d:
 *   ResNet-50 body: 7x7/2 stem, the 3x3 convolutions of every Bottleneck (stride 1, stride 2, dilated
 *   DC5 stage) + FrozenBatchNorm2d + ReLU    /root/reference/models/backbone_scratch.py:58-68,102-141,156-159
 *   DFormer depth stem: four 3x3/2 convolutions + BatchNorm (eval) + GELU
 *                                            /root/reference/models/dformer_backbone.py:18-71,130-159
 *
 * Layout: activations NCHW fp32, packed.  fp32 in, fp32 accumulate (exact fp32 MFMA), fp32 out.
 * Same conventions as dfx_msda.h (device pointers, caller-owned buffers, enqueue-only, 0 / <0).
 */
#ifndef DFX_CONV_H
#define DFX_CONV_H

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_ACT_NONE 0
#define DFX_ACT_RELU 1
#define DFX_ACT_GELU 2 /* exact (erf) GELU, as nn.GELU() */

/* Direct convolution as an implicit GEMM  Y_n[Co, Ho*Wo] = Wp[Co, Kpad] x im2col(X_n)[Kpad, Ho*Wo]
 * with the gather done while the operand tile is staged into LDS (no im2col buffer exists).
 *
 *   x     [N, Ci, H, W]; x_image_stride = elements between consecutive images (0: packed, Ci*H*W) - a channel
 *                     slice of a wider tensor (the RGB / depth planes of an RGB-D clip) is read in place
 *   wp    [Co, Kpad]  weights re-ordered by the caller so that column k multiplies the input element
 *                     that ktab[k] names; Kpad a multiple of 16 (pad columns: weight 0, tap -1)
 *   ktab  int32[Kpad][2] on the device, for this input geometry: {tap index ky * KW + kx (or -1: padding
 *                     column), byte offset of the tap inside one image = 4 * (ci*H*W + ky*dilation*W + kx*dilation)}
 *   bias  [Co] or NULL
 *   y     [N, Co, Ho, Wo],  y = act(conv + bias[co]);  input pixel of tap (ky, kx) for output (oy, ox)
 *                           is (oy*stride - pad + ky*dilation, ox*stride - pad + kx*dilation), zero outside the map
 *   KH * KW <= 64
 */
int dfx_conv2d_igemm_f32(const float *x, const float *wp, const int *ktab, const float *bias, float *y,
                         int N, int Ci, int H, int W, int Co, int Ho, int Wo, int Kpad, int KH, int KW,
                         int stride, int pad, int dilation, int act, long x_image_stride, void *stream);

/* Direct convolution of FEW input channels (Ci * KH * KW <= 160, Co <= 64, stride 1 or 2, no dilation) from an
 * LDS-resident input tile: the whole receptive field of a 16 x 16 output tile is staged once and the gathered
 * [K x pixels] operand of the fp32 MFMA is read from LDS (csrc/conv_tile.hip) - the 7x7/2 ResNet stem
 * (/root/reference/models/backbone_scratch.py:102-141 -> torchvision resnet conv1) and the first convolution of
 * the DFormer depth stem (/root/reference/models/dformer_backbone.py:18-71).  Operands as dfx_conv2d_igemm_f32
 * (same wp [Co, Kpad] in (ky, kx, ci) order, zero padded; no tap table: the kernel derives it).
 * dfx_conv2d_tile_fits: 1 when a geometry is covered. */
int dfx_conv2d_tile_fits(int Ci, int Co, int KH, int KW, int stride, int dilation);
int dfx_conv2d_tile_f32(const float *x, const float *wp, const float *bias, float *y,
                        int N, int Ci, int H, int W, int Co, int Ho, int Wo, int Kpad, int KH, int KW,
                        int stride, int pad, int act, long x_image_stride, void *stream);

/* 3x3, stride 1, padding = dilation ("same") convolution by Winograd F(2x2, 3x3) with every stage in one
 * kernel: input tiles are transformed while they are staged into LDS, the 16 element-wise products run as
 * 16 GEMMs [Co x Ci] x [Ci x tiles] on fp32 MFMA, the output transform + bias + activation happen on the
 * way out.  2.25x fewer multiplications than the direct form.  A dilated convolution (the DC5 stage,
 * dilation 2) is the same computation on the dilation x dilation interleaved sub-lattices of the map.
 *
 *   x   [N, Ci, H, W]       Ci a multiple of 8, N*Ci*H*W < 2^30
 *   u   16*Co*Ci floats     transformed weights  U = G g G^T, 16-byte aligned, in the blocked order the kernel
 *                           stages them - opaque to the caller, produced only by dfx_wino_weights_f32;
 *                           Co a multiple of 64
 *   y   [N, Co, H, W]       y = act(conv + bias[co])
 */
int dfx_conv3x3_wino_f32(const float *x, const float *u, const float *bias, float *y,
                         int N, int Ci, int H, int W, int Co, int dilation, int act, void *stream);

/* w [Co, Ci, 3, 3] (optionally scaled per output channel by scale[Co], the folded FrozenBatchNorm2d
 * weight * rsqrt(var + eps)) -> u, 16*Co*Ci floats in the kernel's staging order
 * [Co/64][Ci/8][16 positions][2][64 co][4 ci];  Ci a multiple of 8, Co of 64 */
int dfx_wino_weights_f32(const float *w, const float *scale, float *u, int Co, int Ci, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_CONV_H */
