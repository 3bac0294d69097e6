// RoIAlign (average pooling, fixed sampling grid) for gfx950.
//
// Algorithm = mmcv 1.7.0 roi_align (avg, aligned): roi scaled by spatial_scale and shifted by
// -0.5 when aligned; each of ph x pw bins averages sampling_ratio^2 bilinear samples; a sample
// outside [-1, size] contributes 0; coordinates are clamped to [0, size-1] before interpolation.
//
// NHWC kernel: one wave per (roi, bin); lane = 4 channels, so every corner fetch of a sample is
// one fully coalesced 16-byte-per-lane read of the pixel's channel vector (1 KiB for C = 256) and
// the bin's output row leaves as one coalesced store.  The encoder memory is already token-major
// ([H*W, C]), so the reference's permute+contiguous copy to NCHW (multi_plusplus.py:498,513)
// disappears.  NCHW kernel: one thread per output element, for API parity with mmcv's layout.
#include "dfx_common.h"
#include "dfx_roi.h"

namespace {

struct Sample {
    int yl, yh, xl, xh;
    float w1, w2, w3, w4;
    bool ok;
};

__device__ __forceinline__ Sample locate(float y, float x, int H, int W)
{
    Sample s;
    s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)fminf(y, (float)H), xl = (int)fminf(x, (float)W);
    int yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
    s.yl = yl; s.yh = yh; s.xl = xl; s.xh = xh;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    return s;
}

struct Roi {
    int b;
    float x1, y1, bin_w, bin_h;
};

__device__ __forceinline__ Roi load_roi(const float *r, float scale, int aligned, int ph, int pw)
{
    Roi o;
    const float off = aligned ? 0.5f : 0.f;
    o.b = (int)r[0];
    o.x1 = r[1] * scale - off;
    o.y1 = r[2] * scale - off;
    float rw = r[3] * scale - off - o.x1, rh = r[4] * scale - off - o.y1;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    o.bin_w = rw / (float)pw;
    o.bin_h = rh / (float)ph;
    return o;
}

__global__ __launch_bounds__(256) void roi_align_nhwc(const float *__restrict__ in, const float *__restrict__ rois,
                                                      int N, int C, int H, int W, long nbins, int ph, int pw,
                                                      float scale, int sr, int aligned, float *__restrict__ out)
{
    const long bin = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per (roi, bin)
    if (bin >= nbins) return;
    const int lane = threadIdx.x & 63;
    const int k = (int)(bin / (ph * pw)), ij = (int)(bin % (ph * pw));
    const int i = ij / pw, j = ij % pw;
    const Roi r = load_roi(rois + 5 * (long)k, scale, aligned, ph, pw);
    const bool live = r.b >= 0 && r.b < N;
    const float *src = in + (long)(live ? r.b : 0) * H * W * C;
    const float inv = 1.f / (float)(sr * sr);
    for (int c = lane * 4; c < C; c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < sr; ++iy) {
            const float y = r.y1 + i * r.bin_h + (iy + 0.5f) * r.bin_h / (float)sr;
            for (int ix = 0; ix < sr; ++ix) {
                const float x = r.x1 + j * r.bin_w + (ix + 0.5f) * r.bin_w / (float)sr;
                const Sample s = locate(y, x, H, W);
                if (!s.ok || !live) continue;
                const float4 a = *reinterpret_cast<const float4 *>(src + ((long)s.yl * W + s.xl) * C + c);
                const float4 b = *reinterpret_cast<const float4 *>(src + ((long)s.yl * W + s.xh) * C + c);
                const float4 d = *reinterpret_cast<const float4 *>(src + ((long)s.yh * W + s.xl) * C + c);
                const float4 e = *reinterpret_cast<const float4 *>(src + ((long)s.yh * W + s.xh) * C + c);
                acc.x += s.w1 * a.x + s.w2 * b.x + s.w3 * d.x + s.w4 * e.x;
                acc.y += s.w1 * a.y + s.w2 * b.y + s.w3 * d.y + s.w4 * e.y;
                acc.z += s.w1 * a.z + s.w2 * b.z + s.w3 * d.z + s.w4 * e.z;
                acc.w += s.w1 * a.w + s.w2 * b.w + s.w3 * d.w + s.w4 * e.w;
            }
        }
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
        *reinterpret_cast<float4 *>(out + bin * C + c) = acc;
    }
}

__global__ __launch_bounds__(256) void roi_align_nchw(const float *__restrict__ in, const float *__restrict__ rois,
                                                      int N, int C, int H, int W, long total, int ph, int pw,
                                                      float scale, int sr, int aligned, float *__restrict__ out)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % pw), i = (int)((idx / pw) % ph);
        const int c = (int)((idx / ((long)pw * ph)) % C), k = (int)(idx / ((long)pw * ph * C));
        const Roi r = load_roi(rois + 5 * (long)k, scale, aligned, ph, pw);
        float acc = 0.f;
        if (r.b >= 0 && r.b < N) {
            const float *src = in + ((long)r.b * C + c) * H * W;
            for (int iy = 0; iy < sr; ++iy) {
                const float y = r.y1 + i * r.bin_h + (iy + 0.5f) * r.bin_h / (float)sr;
                for (int ix = 0; ix < sr; ++ix) {
                    const float x = r.x1 + j * r.bin_w + (ix + 0.5f) * r.bin_w / (float)sr;
                    const Sample s = locate(y, x, H, W);
                    if (!s.ok) continue;
                    acc += s.w1 * src[s.yl * W + s.xl] + s.w2 * src[s.yl * W + s.xh] +
                           s.w3 * src[s.yh * W + s.xl] + s.w4 * src[s.yh * W + s.xh];
                }
            }
        }
        out[idx] = acc / (float)(sr * sr);
    }
}

int check(const void *in, const void *rois, const void *out, int N, int C, int H, int W, int K, int ph, int pw, int sr)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K < 0 || ph <= 0 || pw <= 0)
        return dfx::fail(DFX_EINVAL, "roi_align: bad dimension");
    if (sr <= 0) return dfx::fail(DFX_EINVAL, "roi_align: sampling_ratio must be > 0 (adaptive grids are not used on this path)");
    if (K == 0) return 1;
    if (!in || !rois || !out) return dfx::fail(DFX_EINVAL, "roi_align: null pointer");
    return 0;
}

}  // namespace

extern "C" int dfx_roi_align_nhwc_f32(const float *input, const float *rois, int N, int C, int H, int W, int K,
                                      int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                      float *out, void *stream)
{
    const int rc = check(input, rois, out, N, C, H, W, K, ph, pw, sampling_ratio);
    if (rc < 0) return rc;
    if (rc == 1) return DFX_OK;
    if ((C & 3) || !dfx::aligned16(input) || !dfx::aligned16(out))
        return dfx::fail(DFX_EINVAL, "roi_align nhwc: C %% 4 == 0 and 16-byte aligned buffers required");
    const long nbins = (long)K * ph * pw;
    hipLaunchKernelGGL(roi_align_nhwc, dim3((unsigned)((nbins + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       input, rois, N, C, H, W, nbins, ph, pw, spatial_scale, sampling_ratio, aligned, out);
    return dfx::check_launch("roi_align_nhwc");
}

extern "C" int dfx_roi_align_nchw_f32(const float *input, const float *rois, int N, int C, int H, int W, int K,
                                      int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                                      float *out, void *stream)
{
    const int rc = check(input, rois, out, N, C, H, W, K, ph, pw, sampling_ratio);
    if (rc < 0) return rc;
    if (rc == 1) return DFX_OK;
    const long total = (long)K * C * ph * pw;
    hipLaunchKernelGGL(roi_align_nchw, dim3(dfx::grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       input, rois, N, C, H, W, total, ph, pw, spatial_scale, sampling_ratio, aligned, out);
    return dfx::check_launch("roi_align_nchw");
}
