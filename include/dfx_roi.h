/*
 * dfx_roi.h -- C ABI of the RoIAlign used by the TransVOD++ query/RoI fusion stage.
 *
 * Replaces the third-party op the reference calls on this path:
 *   mmcv.ops.RoIAlign(output_size=7, spatial_scale=1/32, sampling_ratio=2)   (mmcv-full 1.7.0,
 *   pool_mode='avg', aligned=True), constructed at
 *   /root/reference/models/deformable_transformer_multi_plusplus.py:129-132 and called at :499,:514.
 * mmcv is not vendored in the reference and holds no test there: parity is pinned to this
 * repository's own restatement of the published algorithm (oracle/msda_oracle.c).
 *
 * Same conventions as dfx_msda.h: device pointers, caller-owned buffers, enqueue-only on
 * `stream`, 0 / negative return code.
 *   rois [K,5] = (batch index, x1, y1, x2, y2) in input-image pixels (scaled by spatial_scale).
 */
#ifndef DFX_ROI_H
#define DFX_ROI_H

#ifdef __cplusplus
extern "C" {
#endif

/* input [N,C,H,W] -> out [K,C,ph,pw]  (the layout mmcv works in) */
int dfx_roi_align_nchw_f32(const float *input, const float *rois, int N, int C, int H, int W, int K,
                           int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                           float *out, void *stream);

/* input [N,H,W,C] (token-major encoder memory, no transpose needed) -> out [K,ph*pw,C];
 * C must be a multiple of 4 and both buffers 16-byte aligned. */
int dfx_roi_align_nhwc_f32(const float *input, const float *rois, int N, int C, int H, int W, int K,
                           int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                           float *out, void *stream);

/* DynamicConv of the query/RoI fusion head (Sparse R-CNN instance interaction) between the parameter
 * Linear and the output Linear, one launch (/root/reference/models/sparse_roi_head/head.py:98-113):
 *   k1 = params[r, 0 : C*dd] as [C,dd];  k2 = params[r, C*dd : 2*C*dd] as [dd,C]
 *   y  = relu(LayerNorm_dd(feats[r] @ k1));   out[r] = relu(LayerNorm_C(y @ k2))
 * for every RoI r: feats / out [K,R,C] (R = ph*pw <= 64 rows), params [K, >= 2*C*dd] with row stride
 * p_stride (the dynamic_layer output as it stands), C = 256, dd = 64, eps 1e-5 inside the sqrt.
 * The library runs this as two batched GEMMs of 49-row matrices plus four normalisation / ReLU passes
 * over [K,R,C]; here each RoI's matrices stay in one CU's LDS between the two fp32-MFMA products. */
int dfx_dynamic_conv_f32(const float *feats, const float *params, long p_stride,
                         const float *g1, const float *b1, const float *g2, const float *b2,
                         float *out, int K, int R, int C, int dd, float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFX_ROI_H */
