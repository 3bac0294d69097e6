/*
 * dfx_mha.h -- C ABI of the fused scaled-dot-product attention of the 300-query layers (gfx950).
 *
 * It stands where the reference calls torch.nn.MultiheadAttention on this path (always without masks,
 * dropout off in eval): the object-query self-attention of every decoder layer
 *   /root/reference/models/deformable_transformer_single.py:650-655 (DeformableTransformerDecoderLayer)
 * the TQE self- and cross-attention and the TDTD self-attention of the temporal stage
 *   /root/reference/models/deformable_transformer_multi_plusplus.py:815-838, 879-886
 * and the RCNNHead self-attention of the query/RoI fusion
 *   /root/reference/models/sparse_roi_head/head.py:70-76.
 * The in/out projections stay GEMMs (dfx_gemm_f32); this call is the part in between, which the
 * library path runs as  q*scale, two transposing copies, bmm, softmax, bmm, copy  (7-9 launches):
 *
 *   out[b,i,h*32+d] = sum_j softmax_j( scale * <q[b,i,h,:], k[b,j,h,:]> ) * v[b,j,h,d]
 *
 * fp32 (exact-fp32 MFMA, online softmax), head dimension 32, `heads` heads side by side in the last
 * dimension (E = 32*heads).  q [B,Lq,E], k / v [B,Lk,E], out [B,Lq,E]: the last dimension is
 * contiguous, batch and sequence strides are given in floats (multiples of 4), so q / k / v may be
 * column slices of one joint projection output.  Same conventions as dfx_msda.h (device pointers,
 * enqueue-only, 0 / <0).
 */
#ifndef DFX_MHA_H
#define DFX_MHA_H

#ifdef __cplusplus
extern "C" {
#endif

int dfx_mha_f32(const float *q, long q_batch, long q_row,
                const float *k, long k_batch, long k_row,
                const float *v, long v_batch, long v_row,
                float *out, long o_batch, long o_row,
                int B, int heads, int Lq, int Lk, float scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_MHA_H */
