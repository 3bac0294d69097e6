// GroupNorm of the detector's input projections (Conv1x1 + GroupNorm(32, 256),
// /root/reference/models/deformable_detr_single.py:101-125,143-150) as two streaming kernels
// (include/dfx_fused.h, dfx_group_norm_f32):
//   1. statistics: one workgroup per (image, group) - its C/G channels x H*W block of the NCHW map is
//      contiguous - mean, then the centred second moment on a second read (the block is L2-resident),
//      wave shuffles + one LDS exchange for the reductions
//   2. normalise + affine, written either NCHW or token-major [N, H*W, C] through a 64 x 64 LDS
//      transpose (reads run along pixels, writes along channels): the transformer consumes tokens, so the
//      separate flatten/transpose copy of the reference's formulation disappears.
// HBM-bound: the map is read twice (second read from L2 / Infinity Cache) and written once.
#include "dfx_common.h"
#include "dfx_fused.h"

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += red[w];
    return t;
}

// x block of `count` contiguous floats per (n, g); stats[(n*G + g)*2] = mean, [..+1] = rstd
__global__ __launch_bounds__(256) void gn_stats(const float *__restrict__ x, float *__restrict__ stats, long count, float eps)
{
    __shared__ float red[4];
    const float *p = x + (long)blockIdx.x * count;
    const bool vec = (count & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0);
    float s = 0.f;
    if (vec) {
        for (long i = threadIdx.x * 4L; i < count; i += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (long i = threadIdx.x; i < count; i += 256) s += p[i];
    }
    const float mean = block_sum(s, red) / (float)count;
    float q = 0.f;
    if (vec) {
        for (long i = threadIdx.x * 4L; i < count; i += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (long i = threadIdx.x; i < count; i += 256) { const float a = p[i] - mean; q += a * a; }
    }
    const float var = block_sum(q, red) / (float)count;
    if (threadIdx.x == 0) {
        stats[blockIdx.x * 2L] = mean;
        stats[blockIdx.x * 2L + 1] = rsqrtf(var + eps);
    }
}

// y[n][p][c] (tokens) or y[n][c][p] = (x[n][c][p] - mean) * rstd * gamma[c] + beta[c]; tile = 64 channels x 64 pixels
template <bool TOKENS>
__global__ __launch_bounds__(256) void gn_apply(const float *__restrict__ x, const float *__restrict__ stats,
                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                float *__restrict__ y, int C, int HW, int cg)
{
    __shared__ float tile[64][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 4 rows of 64 per pass
    const float *xn = x + (long)n * C * HW;
    float *yn = y + (long)n * C * HW;
    const int G = C / cg;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) {
            const float *st = stats + ((long)n * G + c / cg) * 2;
            const float a = st[1] * gamma[c], b = beta[c] - st[0] * a;
            const float v = xn[(long)c * HW + p] * a + b;
            if (TOKENS) tile[r][tx] = v;
            else yn[(long)c * HW + p] = v;
        }
    }
    if (!TOKENS) return;
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) yn[(long)p * C + c] = tile[tx][r];
    }
}

}  // namespace

extern "C" int dfx_group_norm_f32(const float *x, const float *gamma, const float *beta, float *stats, float *y, int N,
                                  int C, long HW, int groups, float eps, int tokens_out, void *stream)
{
    if (N < 0 || C <= 0 || HW < 0 || groups <= 0 || C % groups) return dfx::fail(DFX_EINVAL, "group_norm: bad dimension");
    if ((long)N * HW == 0) return DFX_OK;
    if (!x || !gamma || !beta || !stats || !y) return dfx::fail(DFX_EINVAL, "group_norm: null pointer");
    if (HW >= (1L << 31) || (long)N * groups >= (1L << 31) || N > 65535) return dfx::fail(DFX_ERANGE, "group_norm: too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cg = C / groups;
    hipLaunchKernelGGL(gn_stats, dim3((unsigned)(N * groups)), dim3(256), 0, st, x, stats, (long)cg * HW, eps);
    int rc = dfx::check_launch("gn_stats");
    if (rc != DFX_OK) return rc;
    const dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N);
    if (tokens_out) hipLaunchKernelGGL(gn_apply<true>, grid, dim3(256), 0, st, x, stats, gamma, beta, y, C, (int)HW, cg);
    else hipLaunchKernelGGL(gn_apply<false>, grid, dim3(256), 0, st, x, stats, gamma, beta, y, C, (int)HW, cg);
    return dfx::check_launch("gn_apply");
}
