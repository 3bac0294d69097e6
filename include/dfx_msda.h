/*
 * dfx_msda.h -- C ABI of the MI355X (gfx950) multi-scale deformable attention path.
 *
 * This is the drop-in boundary: the entry points are exactly what a binding for
 * the reference's only native component would call.  The reference exposes that
 * component as the pybind11 module `MultiScaleDeformableAttention` with
 *   ms_deform_attn_forward   /root/reference/models/ops/src/ms_deform_attn.h:20-38
 *   ms_deform_attn_backward  /root/reference/models/ops/src/ms_deform_attn.h:41-61
 * (bound in models/ops/src/vision.cpp:13-16, implemented for CUDA in
 *  models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153).  The Python shim that
 * turns these C calls back into that module lives in
 *   depth-fusion-in-transformer-based-video-object-detection_amd/MultiScaleDeformableAttention.py
 * and INTEGRATION.md shows the few lines a maintainer of the reference adds.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; no torch / ATen types cross this boundary;
 *   - every pointer is DEVICE memory (HBM) of the current HIP device, including
 *     `shapes` and `lsi` (the reference keeps them on the device as int64);
 *   - buffers are packed, row-major, and addressed FLAT with strides derived
 *     from the integer arguments (ms_deform_attn_cuda.cu:40-48,58-60): a
 *     sampling-location buffer that is larger than N*Lq*M*L*P*2 is legal and
 *     only its prefix is read (the TransVOD temporal decoder relies on this);
 *   - the call only ENQUEUES work on `stream` (a hipStream_t; NULL = the
 *     default stream) and returns; no allocation, no synchronisation, no global
 *     state, re-entrant;
 *   - the caller owns every buffer; outputs are fully overwritten by forward;
 *     backward ACCUMULATES (atomic adds) into grad_value, grad_loc and grad_aw,
 *     so the caller zero-fills all three first (the reference allocates them
 *     with zeros_like, ms_deform_attn_cuda.cu:116-118);
 *   - return value: 0 on success, a negative DFX_E* code otherwise; no C++
 *     exception ever crosses the boundary; dfx_last_error() gives the text.
 */
#ifndef DFX_MSDA_H
#define DFX_MSDA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_OK 0
#define DFX_EINVAL (-1)  /* bad dimension / null pointer                         */
#define DFX_ELAUNCH (-2) /* hipGetLastError() reported a launch failure          */
#define DFX_ERANGE (-3)  /* sizes overflow the 32-bit index space of the kernels */

/* ABI version: bumped whenever a signature below changes. */
int dfx_abi_version(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char *dfx_last_error(void);

/*
 * Measurement aid (bench.py); not part of the reference's surface and the one place with process-wide
 * state.  While enabled, every fused MSDA kernel, every GEMM and every convolution launch is dispatched with
 * hipExtLaunchKernelGGL, which stamps an event pair with the kernel's own begin / end timestamps (the kernel
 * duration proper, on the launch stream, without the dispatch gap of hipEventRecord pairs).
 * dfx_profile_drain waits for the recorded kernels, writes up to `cap` records, frees the events and returns
 * the number written.  Record = (duration in ms, work, tag_a, tag_b):
 *   MSDA kernels      work = algorithmic bytes of the launch, tag_a = Lq (> 0), tag_b = S
 *   gemm_f32_kernel   work = 2*M*N*K*batch flops, tag_a = -1 ([K,N] operand: 1x1 convolution) / -2 (Linear), tag_b = tile
 *   conv_wino_kernel  work = MFMA flops executed (2*16*Co*Ci per 2x2 tile), tag_a = -3, tag_b = dilation
 *   conv_igemm_kernel work = 2*Co*Kpad*Ho*Wo*N flops, tag_a = -4, tag_b = tile rows
 */
int dfx_profile_enable(int on);
int dfx_profile_drain(float *ms, long *bytes, int *lq, int *s, int cap);

/*
 * The launchers' DFX_* tuning / diagnostic environment switches are read once per process;
 * this re-reads them (A/B tools and tests that change the environment of a running process).
 */
int dfx_tuning_reload(void);

/*
 * Forward: out[b,q,m,c] = sum_{l,p} aw[b,q,m,l,p] * bilinear(value_l[b,:,m,c], x*W_l-0.5, y*H_l-0.5)
 * Replaces ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:20-80) and the
 * kernel ms_deformable_im2col_gpu_kernel (ms_deform_im2col_cuda.cuh:237-299).
 *   value  [N,S,M,D]        shapes int64 [L,2] = (H_l, W_l)      lsi int64 [L]
 *   loc    >= N*Lq*M*L*P*2  aw >= N*Lq*M*L*P                     out [N,Lq,M*D]
 */
int dfx_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *aw,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         float *out, void *stream);
int dfx_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                         const double *loc, const double *aw,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         double *out, void *stream);

/*
 * Backward.  Replaces ms_deform_attn_cuda_backward (ms_deform_attn_cuda.cu:83-153)
 * and the six col2im kernel variants (ms_deform_im2col_cuda.cuh:301-920).
 *   grad_out [N,Lq,M*D]; grad_value like value; grad_loc like loc; grad_aw like aw.
 */
int dfx_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *aw, const float *grad_out,
                          int N, int S, int M, int D, int L, int Lq, int P,
                          float *grad_value, float *grad_loc, float *grad_aw, void *stream);
int dfx_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                          const double *loc, const double *aw, const double *grad_out,
                          int N, int S, int M, int D, int L, int Lq, int P,
                          double *grad_value, double *grad_loc, double *grad_aw, void *stream);

/*
 * Fused front end + sampling (inference): what MSDeformAttn.forward does between
 * its Linear layers (/root/reference/models/ops/modules/ms_deform_attn.py:98-114):
 *   aw  = softmax over (L*P) of logits[b,q,m,:]
 *   loc = ref[b,q,l,:2] + off[b,q,m,l,p,:] / (W_l,H_l)                (ref_dim == 2)
 *   loc = ref[b,q,l,:2] + off[b,q,m,l,p,:] / P * ref[b,q,l,2:] * 0.5  (ref_dim == 4)
 * followed by the forward above, without materialising loc / aw in HBM.
 *   off    [N,Lq,M,L,P,2] raw sampling_offsets Linear output, row stride
 *          `off_stride` floats between consecutive (b,q) rows (>= M*L*P*2)
 *   logits [N,Lq,M,L*P]   raw attention_weights Linear output, row stride
 *          `logit_stride` floats (>= M*L*P); both strides let one GEMM write a
 *          concatenated [offsets | logits] row
 *   ref    [N,Lq,Lr,ref_dim]; Lr == L normally.  Lr > L reproduces the flat
 *          read of the TransVOD temporal decoder: the
 *          virtual location tensor [Lq,M,Lr,P,2] of EACH batch element is read
 *          flat with L levels (the reference only does this with N == 1; for
 *          N > 1 every batch element follows the N == 1 rule on its own).
 */
int dfx_msda_fused_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                               const float *ref, int ref_dim, int Lr,
                               const float *off, long off_stride,
                               const float *logits, long logit_stride,
                               int N, int S, int M, int D, int L, int Lq, int P,
                               float *out, void *stream);

/*
 * The same fused operator for single-level attention with the whole level resident in LDS
 * (csrc/msda_level.hip): L = 1, one reference level, P = 4, M = 8, D = 32, any Lq.  A workgroup
 * stages 8 channels of one head of every token of the H x W map (zero-bordered) into LDS and the
 * bilinear gathers never leave the CU.  H and W are HOST integers (the LDS image is sized from
 * them); the level must fit: dfx_msda_fused_level_fits(H, W) != 0, i.e. (H+3)*(W+2)*32 B within
 * the CU's 160 KB (50 x 84 at 800 x 1333 does).  Results equal dfx_msda_fused_forward_f32 up to
 * rounding (softmax and the location quotient are computed with correctly rounded divisions, the
 * exponential to 1 ulp).  Replaces the same reference lines as dfx_msda_fused_forward_f32
 * (models/ops/modules/ms_deform_attn.py:98-114).
 *
 * Operands are addressed through dfx_msda_level_layout (strides in floats, multiples of 4):
 *   value   channel c of head m of token s of frame n at
 *           value[n*value_frame + s*value_token + m*value_head + (c/8)*value_oct + ((c/4)%2)*value_chunk + c%4]
 *           reference layout [N,S,8,32]: (S*256, 256, 32, 8, 4); the layout the kernel is built for is
 *           dfx_gemm_f32's c_block = 4 output [64][N*S][4]: (S*4, 4, 32*N*S, 8*N*S, 4*N*S), where every
 *           16-byte chunk plane a workgroup stages is contiguous
 *   off     the 4 (x, y) offsets of head m of query row r = n*Lq + q at off[r*off_row + m*off_head + 0..7]
 *   logits  its 4 attention logits at logits[r*logit_row + m*logit_head + 0..3]
 *           reference Linear outputs: off (64-float rows, head stride 8), logits (32-float rows, head
 *           stride 4); built-for layout: one c_block = 12 GEMM output [8][N*Lq][12] holding
 *           (offsets, logits) of a head side by side: off_row = logit_row = 12,
 *           off_head = logit_head = N*Lq*12, logits = off + 8
 *   out     channel c of head m of query row r at
 *           out[r*out_row + m*out_head + (c/8)*out_oct + ((c/4)%2)*out_chunk + c%4]
 *           reference layout [N,Lq,256]: (256, 32, 8, 4); built-for layout [64][N*Lq][4]
 *           (dfx_gemm_f32's K-block-major A operand of output_proj): (4, 32*N*Lq, 8*N*Lq, 4*N*Lq)
 *   ref [N,Lq,1,ref_dim]
 */
typedef struct dfx_msda_level_layout {
    long value_frame, value_token, value_head, value_oct, value_chunk;
    long off_row, off_head, logit_row, logit_head;
    long out_row, out_head, out_oct, out_chunk;
} dfx_msda_level_layout;

int dfx_msda_fused_level_fits(int H, int W);
int dfx_msda_fused_level_forward_f32(const float *value, const float *ref, int ref_dim,
                                     const float *off, const float *logits,
                                     const dfx_msda_level_layout *layout,
                                     int N, int H, int W, int Lq, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_MSDA_H */
