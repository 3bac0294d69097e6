// Fused input preprocessing for gfx950: Pillow-exact bilinear resize + /255 + normalise + pad + mask
// (include/dfx_preprocess.h).  One thread per output pixel of the padded frame; the horizontal taps
// of each needed source row are reduced to the uint8 value Pillow's horizontal pass would have stored,
// then the vertical taps are applied - same integer arithmetic, so results are bit-identical to
// PIL.Image.resize(BILINEAR) followed by ToTensor + Normalize.  HBM-bound: the source image is read
// ~support^2 times through L1/L2, the fp32 output written once.
#include "dfx_common.h"
#include "dfx_preprocess.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v)
{
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void preprocess_u8(const uint8_t *__restrict__ src, int Hs, int Ws, int Cs,
                                                     const int32_t *__restrict__ xb, const int32_t *__restrict__ xk, int kx,
                                                     const int32_t *__restrict__ yb, const int32_t *__restrict__ yk, int ky,
                                                     int Ho, int Wo, const float *__restrict__ mean,
                                                     const float *__restrict__ stdv, float *__restrict__ dst,
                                                     long plane_stride, int Hp, int Wp, uint8_t *__restrict__ mask)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Hp * Wp) return;
    const int y = (int)(idx / Wp), x = (int)(idx % Wp);
    const bool inside = y < Ho && x < Wo;
    if (mask) mask[idx] = inside ? 0 : 1;
    if (!inside) {
        for (int c = 0; c < Cs; ++c) dst[c * plane_stride + idx] = 0.f;
        return;
    }
    const int xmin = kx ? xb[2 * x] : x, xn = kx ? xb[2 * x + 1] : 1;
    const int ymin = ky ? yb[2 * y] : y, yn = ky ? yb[2 * y + 1] : 1;
    for (int c = 0; c < Cs; ++c) {
        int accv = 1 << (PRECISION_BITS - 1);
        int last = 0;
        for (int r = 0; r < yn; ++r) {
            const uint8_t *row = src + ((long)(ymin + r) * Ws + xmin) * Cs + c;
            int h;
            if (kx) {
                int acch = 1 << (PRECISION_BITS - 1);
                for (int t = 0; t < xn; ++t) acch += (int)row[(long)t * Cs] * xk[(long)x * kx + t];
                h = clip8(acch);
            } else {
                h = row[0];
            }
            last = h;
            if (ky) accv += h * yk[(long)y * ky + r];
        }
        const int v = ky ? clip8(accv) : last;
        dst[c * plane_stride + idx] = ((float)v / 255.f - mean[c]) / stdv[c];
    }
}

}  // namespace

extern "C" int dfx_preprocess_u8_f32(const uint8_t *src, int Hs, int Ws, int Cs, const int32_t *xbounds,
                                     const int32_t *xcoef, int kx, const int32_t *ybounds, const int32_t *ycoef, int ky,
                                     int Ho, int Wo, const float *mean, const float *std, float *dst, long plane_stride,
                                     int Hp, int Wp, uint8_t *mask, void *stream)
{
    if (Hs <= 0 || Ws <= 0 || Cs <= 0 || Cs > 4 || Ho <= 0 || Wo <= 0 || Hp < Ho || Wp < Wo || kx < 0 || ky < 0)
        return dfx::fail(DFX_EINVAL, "preprocess: bad dimension");
    if (!src || !mean || !std || !dst || (kx && (!xbounds || !xcoef)) || (ky && (!ybounds || !ycoef)))
        return dfx::fail(DFX_EINVAL, "preprocess: null pointer");
    if ((kx == 0 && Wo != Ws) || (ky == 0 && Ho != Hs))
        return dfx::fail(DFX_EINVAL, "preprocess: a pass can only be skipped when the size is unchanged");
    const long total = (long)Hp * Wp;
    hipLaunchKernelGGL(preprocess_u8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, Hs, Ws, Cs, xbounds, xcoef, kx, ybounds, ycoef, ky, Ho, Wo, mean, std, dst, plane_stride, Hp,
                       Wp, mask);
    return dfx::check_launch("preprocess_u8");
}
