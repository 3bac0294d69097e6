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

// residual add + LayerNorm: one wave per row, the row lives in registers (<= 4 float4 per lane),
// mean and variance by two in-register passes + xor-shuffle reductions (same formula as
// nn.LayerNorm: biased variance, eps under the square root).
template <int CHUNKS>
__global__ __launch_bounds__(256) void add_layernorm(const float *__restrict__ x, const float *__restrict__ res,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     float *__restrict__ out, long rows, int C, float eps)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float4 v[CHUNKS];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        const int c = k * 256 + lane * 4;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            v[k] = *reinterpret_cast<const float4 *>(x + row * C + c);
            if (res) {
                const float4 r = *reinterpret_cast<const float4 *>(res + row * C + c);
                v[k].x += r.x; v[k].y += r.y; v[k].z += r.z; v[k].w += r.w;
            }
            sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        if (k * 256 + lane * 4 < C) {
            const float a = v[k].x - mean, b = v[k].y - mean, c2 = v[k].z - mean, d = v[k].w - mean;
            sq += (a * a + b * b) + (c2 * c2 + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        const int c = k * 256 + lane * 4;
        if (c < C) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
            const float4 b = *reinterpret_cast<const float4 *>(beta + c);
            float4 o;
            o.x = (v[k].x - mean) * rstd * g.x + b.x;
            o.y = (v[k].y - mean) * rstd * g.y + b.y;
            o.z = (v[k].z - mean) * rstd * g.z + b.z;
            o.w = (v[k].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<float4 *>(out + row * C + c) = o;
        }
    }
}

// stem epilogue: relu and +bias are monotonic, so relu(max(window) + b) == max(relu(window + b)).
// One output row of one (image, channel) plane per workgroup pass; a thread makes 4 neighbouring outputs
// from input columns 2*ox-1 .. 2*ox+7 of the 3 rows: per row one scalar and two 16-byte loads (rows of an
// odd width are only 4-byte aligned, which global dwordx4 loads accept) instead of 9 scalar loads per output.
__global__ __launch_bounds__(256) void bias_relu_maxpool(const float *__restrict__ x, const float *__restrict__ bias,
                                                         float *__restrict__ out, int C, int H, int W, int Ho, int Wo,
                                                         long rows)
{
    // items = (output row, group of 4 output columns), one per thread, so that every lane of a workgroup works whatever
    // the row width (a 334-wide row has 84 groups: a workgroup per row left two thirds of its lanes idle)
    const int Q = (Wo + 3) / 4;
    const long items = rows * Q;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long row = it / Q;
        const int ox = (int)(it - row * Q) * 4;
        const int oy = (int)(row % Ho);
        const long plane = row / Ho;
        const float b = bias[plane % C];
        const float *src = x + plane * H * W;
        const int y0 = oy * 2 - 1;
        const int xl = ox * 2 - 1;                       // leftmost input column (may be -1)
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y0 + dy;
            if (yy < 0 || yy >= H) continue;
            const float *r = src + (long)yy * W;
            float v[9];
            v[0] = xl >= 0 ? r[xl] : -INFINITY;
            if (xl + 8 < W) {                            // columns xl+1 .. xl+8 all inside the row
                const float4 p = *reinterpret_cast<const float4 *>(r + xl + 1);
                const float4 q = *reinterpret_cast<const float4 *>(r + xl + 5);
                v[1] = p.x; v[2] = p.y; v[3] = p.z; v[4] = p.w; v[5] = q.x; v[6] = q.y; v[7] = q.z; v[8] = q.w;
            } else {
#pragma unroll
                for (int k = 1; k < 9; ++k) v[k] = xl + k < W ? r[xl + k] : -INFINITY;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) m[o] = fmaxf(m[o], fmaxf(fmaxf(v[2 * o], v[2 * o + 1]), v[2 * o + 2]));
        }
        float *dst = out + row * Wo + ox;
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (ox + o < Wo) dst[o] = fmaxf(m[o] + b, 0.f);
    }
}

}  // namespace

extern "C" int dfx_bias_relu_maxpool_f32(const float *x, const float *bias, float *out, int N, int C, int H, int W,
                                         void *stream)
{
    if (N < 0 || C <= 0 || H <= 0 || W <= 0) return dfx::fail(DFX_EINVAL, "bias_relu_maxpool: bad dimension");
    if (N == 0) return DFX_OK;
    if (!x || !bias || !out) return dfx::fail(DFX_EINVAL, "bias_relu_maxpool: null pointer");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;          // floor((H + 2 - 3) / 2) + 1
    const long rows = (long)N * C * Ho;
    const long blocks = (rows * ((Wo + 3) / 4) + 255) / 256;
    const unsigned grid = (unsigned)(blocks < (1L << 20) ? blocks : (1L << 20));
    hipLaunchKernelGGL(bias_relu_maxpool, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, bias, out, C, H,
                       W, Ho, Wo, rows);
    return dfx::check_launch("bias_relu_maxpool");
}

// Box refinement of the iterative decoders in one pass (8 elementwise launches in the reference's
// formulation): out = sigmoid(delta + inverse_sigmoid(ref)) on the first ref_dim of the 4 box columns,
// sigmoid(delta) on the rest; inverse_sigmoid as util/misc.py:55-58 (clamp to [0,1], eps = 1e-5 floors).
__global__ __launch_bounds__(256) void box_refine(const float4 *__restrict__ delta, const float *__restrict__ ref,
                                                  int ref_dim, float4 *__restrict__ out, long rows, float eps)
{
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float4 d = delta[r];
    float v[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < ref_dim) {
            float x = ref[r * ref_dim + c];
            x = fminf(fmaxf(x, 0.f), 1.f);
            v[c] += logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
        }
        v[c] = 1.f / (1.f + expf(-v[c]));
    }
    out[r] = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" int dfx_box_refine_f32(const float *delta, const float *ref, int ref_dim, float *out, long rows, float eps,
                                  void *stream)
{
    if (rows < 0 || (ref_dim != 2 && ref_dim != 4)) return dfx::fail(DFX_EINVAL, "box_refine: bad dimension");
    if (rows == 0) return DFX_OK;
    if (!delta || !ref || !out) return dfx::fail(DFX_EINVAL, "box_refine: null pointer");
    if (!dfx::aligned16(delta) || !dfx::aligned16(out)) return dfx::fail(DFX_EINVAL, "box_refine: buffers must be 16-byte aligned");
    if ((rows + 255) / 256 >= (1L << 31)) return dfx::fail(DFX_ERANGE, "box_refine: too many rows");
    hipLaunchKernelGGL(box_refine, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(delta), ref, ref_dim, reinterpret_cast<float4 *>(out), rows, eps);
    return dfx::check_launch("box_refine");
}

extern "C" int dfx_add_layernorm_f32(const float *x, const float *res, const float *gamma, const float *beta,
                                     float *out, long rows, int C, float eps, void *stream)
{
    if (rows < 0 || C <= 0) return dfx::fail(DFX_EINVAL, "add_layernorm: bad dimension");
    if (rows == 0) return DFX_OK;
    if (!x || !gamma || !beta || !out) return dfx::fail(DFX_EINVAL, "add_layernorm: null pointer");
    if ((C & 3) || C > 1024 || !dfx::aligned16(x) || !dfx::aligned16(out) || !dfx::aligned16(gamma) ||
        !dfx::aligned16(beta) || (res && !dfx::aligned16(res)))
        return dfx::fail(DFX_EINVAL, "add_layernorm: C must be a multiple of 4 and <= 1024, buffers 16-byte aligned");
    if ((rows + 3) / 4 >= (1L << 31)) return dfx::fail(DFX_ERANGE, "add_layernorm: too many rows");
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = (C + 255) / 256;
    if (chunks == 1) hipLaunchKernelGGL(add_layernorm<1>, grid, block, 0, st, x, res, gamma, beta, out, rows, C, eps);
    else if (chunks == 2) hipLaunchKernelGGL(add_layernorm<2>, grid, block, 0, st, x, res, gamma, beta, out, rows, C, eps);
    else hipLaunchKernelGGL(add_layernorm<4>, grid, block, 0, st, x, res, gamma, beta, out, rows, C, eps);
    return dfx::check_launch("add_layernorm");
}

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
