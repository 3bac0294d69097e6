/*
 * dfx_gemm.h -- C ABI of the hand-written fp32 MFMA GEMM (gfx950, v_mfma_f32_32x32x2_f32) with the
 * fused prologue / epilogues the deformable-attention path needs.
 *
 * It stands where the reference calls PyTorch library GEMMs on this path:
 *   nn.Linear value_proj / sampling_offsets / attention_weights / output_proj
 *                                   /root/reference/models/ops/modules/ms_deform_attn.py:94-100,116
 *   FFN linear1 (+ReLU) / linear2   /root/reference/models/deformable_transformer_single.py:544-548
 *   Late Fusion adapt / FFN Linears /root/reference/models/deformable_transformer_single.py:379-402
 *   1x1 convolutions of ResNet-50 (+ folded FrozenBatchNorm2d, ReLU, residual add)
 *                                   /root/reference/models/backbone_scratch.py:102-141 (torchvision Bottleneck)
 *   input_proj 1x1 convolution      /root/reference/models/deformable_detr_single.py:101-125
 *
 *   C[b] = act( (A[b] (+ A2[b])) x op(B[b]) + bias (+ R[b]) ), rows with row_mask != 0 forced to 0
 *
 *   A   [M,K] row-major, leading dimension lda; batch stride strideA (0 = shared by all batches)
 *       or, with a_block_stride > 0, K-block-major [K/4][M][4]: element (m, k) at
 *       A[(k / 4) * a_block_stride + m * 4 + k % 4] (the layout the level-in-LDS MSDA kernel writes its
 *       output in, consumed by output_proj; lda is ignored, no A2)
 *   A2  optional, same layout as A, added element-wise while A is staged (query = src + pos)
 *   B   b_is_kn = 1: [K,N] row-major (ldb >= N)  -- activations of a 1x1 convolution, NCHW
 *       b_is_kn = 0: [N,K] row-major (ldb >= K)  -- an nn.Linear weight
 *   bias optional; bias_per_row = 1: bias[m] (convolution), 0: bias[n] (Linear)
 *   R   optional residual, layout of C (ldr, strideR)
 *   row_mask optional uint8[M] per batch (strideMask): value_proj's masked_fill of padded tokens
 *   C   [M,N] row-major, ldc; or, with c_block = w > 0, column-block-major: element (m, n) at
 *       C[(n / w) * c_block_stride + m * w + n % w] - the layout the level-in-LDS MSDA kernel
 *       (dfx_msda.h) reads: value_proj output as [32 channel octets][tokens][8], the joint
 *       sampling_offsets / attention_weights output as [8 heads][queries][12] (no residual then).
 *       A wide block (w a multiple of 128, [N,K] operand) stacks Linears that share A: every block is
 *       a contiguous [M, w] result of its own (the value projections of the six decoder layers over one
 *       memory, /root/reference/models/deformable_transformer_single.py:703-748); they run as one launch
 *       that reads an A panel once for the whole stack
 *   relu 0: none, 1: ReLU, 2: exact (erf) GELU (nn.GELU() of the fusion blocks' FFN,
 *       /root/reference/models/deformable_transformer_single.py:379-402)
 * fp32 in, fp32 accumulate (exact fp32 MFMA), fp32 out.  K must be a multiple of 4; A, B rows
 * 16-byte aligned.  Same conventions as dfx_msda.h (device pointers, enqueue-only, 0 / <0).
 *
 * dfx_gemm_splitk_f32: the same product for problems with few output tiles and a long K (RCNNHead's
 * out_layer 12544 -> 256 over 300 rows per frame, /root/reference/models/sparse_roi_head/head.py:127-172):
 * K is cut into ``splits`` ranges of whole 16-deep steps that run as independent workgroups writing partial
 * [M,N] slabs to ``workspace`` (splits * M * N floats, caller-owned), followed by one reduction launch that
 * adds the slabs, bias, residual and applies the activation.  Row-major A [M,K], C [M,N]; N % 4 == 0.
 *
 * dfx_conv1x1_pair_f32: Y[n] = act(W x [X1[n] ; X2[n]] + bias) - two NCHW inputs of the same map size concatenated
 * along the channels INSIDE the product: the last 1x1 convolution of a bottleneck and the stride-1 projection
 * shortcut of the same block (torchvision Bottleneck: out = relu(bn3(conv3(t)) + downsample(x)),
 * /root/reference/models/backbone_scratch.py:102-141 via resnet50's layer1[0] and the dilated layer4[0]) as ONE GEMM
 * with W = [W3 | Wd] (frozen-BN scales folded in, bias = shift3 + shiftd): the shortcut map is neither written nor
 * read back as a residual.  X1 [batch,K1,HW], X2 [batch,K2,HW] (image strides in floats), W [Co,K1+K2] row-major,
 * Y [batch,Co,HW]; K1, K2 multiples of 16, HW of 4.
 *
 * dfx_linear_ln_f32: C = LayerNorm(R + act((A (+ A2)) x W^T + bias)) over rows of exactly 256 columns (d_model) in ONE
 * launch - the Linear that ends a transformer sub-block with the residual add and the LayerNorm that follow it
 * (output_proj / linear2 + dropout(identity) + norm: /root/reference/models/deformable_transformer_single.py:544-560,
 * 379-402, 596-648).  A workgroup owns whole rows (64 x 256 tile), so the row statistics are wave reductions in the
 * epilogue: mean, then the centred second moment (two passes over registers), eps inside the square root, gamma / beta.
 * act_first = 1 applies the activation before the residual is added (the fusion blocks' LN(t + GELU(Linear(t)))), 0 after.
 * A [M,K] row-major (lda) or K-block-major (a_block_stride > 0, as dfx_gemm_f32), W [256,K] row-major (ldw), R / C [M,256].
 */
#ifndef DFX_GEMM_H
#define DFX_GEMM_H

#ifdef __cplusplus
extern "C" {
#endif

int dfx_gemm_f32(const float *A, const float *A2, long lda, long strideA,
                 const float *B, long ldb, long strideB, int b_is_kn,
                 const float *bias, int bias_per_row,
                 const float *R, long ldr, long strideR,
                 const unsigned char *row_mask, long strideMask,
                 float *C, long ldc, long strideC,
                 int M, int N, int K, int batch, int relu,
                 int c_block, long c_block_stride, long a_block_stride, void *stream);

int dfx_gemm_splitk_f32(const float *A, long lda, const float *B, long ldb, int b_is_kn,
                        const float *bias, int bias_per_row, const float *R, long ldr,
                        float *C, long ldc, int M, int N, int K, int act, int splits,
                        float *workspace, void *stream);

int dfx_linear_ln_f32(const float *A, const float *A2, long lda, long a_block_stride, const float *W, long ldw,
                      const float *bias, const float *R, long ldr, const float *gamma, const float *beta, float eps,
                      float *C, long ldc, int M, int K, int act, int act_first, void *stream);

int dfx_conv1x1_pair_f32(const float *W, const float *X1, long strideX1, int K1,
                         const float *X2, long strideX2, int K2, const float *bias,
                         float *Y, long strideY, int Co, int HW, int batch, int act, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_GEMM_H */
