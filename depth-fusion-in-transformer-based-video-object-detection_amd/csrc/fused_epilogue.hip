// Fused per-channel epilogues of the backbone convolutions (gfx950): one HBM pass instead of the
// reference's mul + add (FrozenBatchNorm2d) + add (residual) + clamp (ReLU) kernels.
// HBM-bound: 4 B read (+4 B residual) + 4 B written per element; 16-byte vector accesses, one
// (image, channel) plane per workgroup row so the bias is a scalar.
#include "dfx_common.h"
#include "dfx_fused.h"

namespace {

template <bool RES, bool RELU, bool VEC>
__global__ __launch_bounds__(256) void bias_act_nchw(const float *__restrict__ x, const float *__restrict__ bias,
                                                     const float *__restrict__ res, float *__restrict__ out,
                                                     int C, long HW, int chunks)
{
    const long plane = blockIdx.x / chunks;           // n * C + c
    const int chunk = blockIdx.x % chunks;
    const float b = bias[plane % C];
    const long base = plane * HW;
    if (VEC) {
        const long n4 = HW >> 2;
        const float4 *xs = reinterpret_cast<const float4 *>(x + base);
        const float4 *rs = reinterpret_cast<const float4 *>(res + base);
        float4 *os = reinterpret_cast<float4 *>(out + base);
        for (long i = (long)chunk * 256 + threadIdx.x; i < n4; i += (long)chunks * 256) {
            float4 v = xs[i];
            v.x += b; v.y += b; v.z += b; v.w += b;
            if (RES) { const float4 r = rs[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            os[i] = v;
        }
    } else {
        for (long i = (long)chunk * 256 + threadIdx.x; i < HW; i += (long)chunks * 256) {
            float v = x[base + i] + b;
            if (RES) v += res[base + i];
            if (RELU) v = fmaxf(v, 0.f);
            out[base + i] = v;
        }
    }
}

}  // namespace

extern "C" int dfx_bias_act_nchw_f32(const float *x, const float *bias, const float *residual, float *out, int N,
                                     int C, long HW, int relu, void *stream)
{
    if (N < 0 || C <= 0 || HW < 0) return dfx::fail(DFX_EINVAL, "bias_act: bad dimension");
    if ((long)N * C * HW == 0) return DFX_OK;
    if (!x || !bias || !out) return dfx::fail(DFX_EINVAL, "bias_act: null pointer");
    const long planes = (long)N * C;
    // split every plane into enough chunks to fill the chip even for few large planes
    int chunks = 1;
    const long per = (HW + 3) / 4;
    while (chunks < 64 && planes * chunks < 4096 && per / (chunks * 2) >= 512) chunks *= 2;
    if (planes * chunks >= (1L << 31)) return dfx::fail(DFX_ERANGE, "bias_act: too many planes");
    const bool vec = (HW % 4 == 0) && dfx::aligned16(x) && dfx::aligned16(out) && (!residual || dfx::aligned16(residual));
    const dim3 grid((unsigned)(planes * chunks)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DFX_GO(RES, RELU, VEC) \
    hipLaunchKernelGGL((bias_act_nchw<RES, RELU, VEC>), grid, block, 0, st, x, bias, residual, out, C, HW, chunks)
    if (residual) {
        if (relu) { if (vec) DFX_GO(true, true, true); else DFX_GO(true, true, false); }
        else      { if (vec) DFX_GO(true, false, true); else DFX_GO(true, false, false); }
    } else {
        if (relu) { if (vec) DFX_GO(false, true, true); else DFX_GO(false, true, false); }
        else      { if (vec) DFX_GO(false, false, true); else DFX_GO(false, false, false); }
    }
#undef DFX_GO
    return dfx::check_launch("bias_act_nchw");
}
