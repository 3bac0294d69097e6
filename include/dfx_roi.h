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

#ifdef __cplusplus
}
#endif
#endif /* DFX_ROI_H */
