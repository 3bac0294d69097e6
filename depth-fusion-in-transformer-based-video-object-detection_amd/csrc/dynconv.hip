// DynamicConv of the TransVOD++ query/RoI fusion head in one launch (include/dfx_roi.h), gfx950.
//
// Per RoI: X [49,256] @ K1 [256,64] -> LayerNorm(64) + ReLU -> @ K2 [64,256] -> LayerNorm(256) + ReLU.
// K1 / K2 are per-RoI (128 KB of parameters each), so the library sees 9600 independent 49-row GEMMs
// and the four elementwise passes stream [K,49,256] through HBM four more times.  Here a persistent
// workgroup (4 waves, one per SIMD) walks the RoIs; a RoI's matrices live in LDS (X 64 x 260, K1 then
// K2 256 x 64 / 64 x 256, Y1 64 x 68 floats = 147 KB) and both products run on fp32 MFMA
// (32 x 32 x 2) with rows padded to 64:
//   product 1: 2 x 2 tiles, one per wave, 128 MFMAs each;   product 2: 2 x 8 tiles, four per wave.
// A operands are read with ds_read_b128 along k (k-permuted MFMA order, as in gemm_f32.hip), B
// operands with ds_read_b32 along n.  The LayerNorms read the tile rows back from LDS, 16 lanes per
// row.  Global loads are issued half a RoI ahead into registers: K2 while product 1 runs, the next
// RoI's X and K1 while product 2 and the last LayerNorm run (one register set serves K1 and K2: each
// is in LDS before the other is loaded).
#include "dfx_common.h"
#include "dfx_roi.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;     // staging registers (plain vectors: arrays of HIP's float4 struct were left in scratch memory)
constexpr int C = 256, DD = 64, RP = 64;       // channels, dynamic dim, padded rows
constexpr int XP = C + 4;                      // X / Y2 row pitch (floats): 65 sixteen-byte slots
constexpr int YP = DD + 4;                     // Y1 row pitch: 17 slots
constexpr int X_F4 = (49 * C / 4 + 255) / 256; // float4 per thread for an X of up to 49 rows (13)
constexpr int X_F4_MAX = RP * C / 4 / 256;     //                                up to 64 rows (16)
constexpr int K_F4 = C * DD / 4 / 256;         // float4 per thread for K1 or K2 (16)

// sum over the 16 lanes of an aligned 16-lane group (all 16 get the result)
__device__ __forceinline__ float sum16(float v)
{
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

template <int XF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dynamic_conv(const float *__restrict__ feats, const float *__restrict__ params,
                                                    long p_stride, const float *__restrict__ g1,
                                                    const float *__restrict__ b1, const float *__restrict__ g2,
                                                    const float *__restrict__ b2, float *__restrict__ out, int K,
                                                    int R, float eps)
{
    extern __shared__ float lds[];
    float *xs = lds;                          // [RP][XP]   X, later Y2 (rows >= R stay zero)
    float *kw = xs + RP * XP;                 // [256][64] K1, later [64][256] K2 (same 16384 floats)
    float *ys = kw + C * DD;                  // [RP][YP]   Y1
    float *lnp = ys + RP * YP;                // g1[64] b1[64] g2[256] b2[256]: LayerNorm parameters (kept out of registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int mt = wave >> 1, nh = wave & 1;

    f32x4 xr[XF], kr[K_F4];                   // staging registers: the next X; K1 or K2
// (macros, not lambdas: with by-reference lambda captures the staging arrays ended up in scratch memory)
#define LOAD_X(ROI)                                                                                              \
    do {                                                                                                         \
        const float *xp_ = feats + (long)(ROI) * R * C;                                                          \
        _Pragma("unroll") for (int u = 0; u < XF; ++u) {                                                         \
            const int e = tid + u * 256, row = e >> 6, c4 = e & 63; /* 64 float4 per row */                      \
            xr[u] = row < R ? *reinterpret_cast<const f32x4 *>(xp_ + row * C + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};       \
        }                                                                                                        \
    } while (0)
#define LOAD_K(PTR)                                                                                              \
    do {                                                                                                         \
        const float *kp_ = (PTR);                                                                                \
        _Pragma("unroll") for (int u = 0; u < K_F4; ++u)                                                         \
            kr[u] = *reinterpret_cast<const f32x4 *>(kp_ + (tid + u * 256) * 4);                                 \
    } while (0)
#define STORE_K()                                                                                                \
    do {                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < K_F4; ++u)                                                         \
            *reinterpret_cast<f32x4 *>(kw + (tid + u * 256) * 4) = kr[u];                                        \
    } while (0)
    for (int e = tid; e < 2 * DD + 2 * C; e += 256)
        lnp[e] = e < DD ? g1[e] : e < 2 * DD ? b1[e - DD] : e < 2 * DD + C ? g2[e - 2 * DD] : b2[e - 2 * DD - C];
    // rows R..63 of X are the zero padding of the 64-row MFMA tiles: written once, never overwritten
    for (int e = tid; e < (RP - R) * (XP / 4); e += 256)
        *reinterpret_cast<float4 *>(xs + R * XP + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);

    int roi = blockIdx.x;
    if (roi < K) {
        LOAD_X(roi);
        LOAD_K(params + (long)roi * p_stride);
    }
    for (; roi < K; roi += gridDim.x) {
#pragma unroll
        for (int u = 0; u < XF; ++u) {
            const int e = tid + u * 256, row = e >> 6, c4 = e & 63;
            if (row < R) *reinterpret_cast<f32x4 *>(xs + row * XP + c4 * 4) = xr[u];
        }
        STORE_K();                                                        // K1
        __syncthreads();
        LOAD_K(params + (long)roi * p_stride + C * DD);                   // K2: in flight during product 1
        // ---- product 1: Y1[64,64] = X[64,256] K1[256,64]; wave = tile (mt, nh) ----
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *arow = xs + (mt * 32 + col) * XP + half * 4;
            const float *bcol = kw + nh * 32 + col + half * 4 * DD;
            float4 a = *reinterpret_cast<const float4 *>(arow);
            float bq[4] = {bcol[0], bcol[DD], bcol[2 * DD], bcol[3 * DD]};
#pragma unroll 2
            for (int j = 0; j < C / 8; ++j) {
                float4 an = a;
                float bn[4] = {bq[0], bq[1], bq[2], bq[3]};
                if (j + 1 < C / 8) {                                      // next group's operands ahead of this group's MFMAs
                    an = *reinterpret_cast<const float4 *>(arow + (j + 1) * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bn[q] = bcol[((j + 1) * 8 + q) * DD];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[3], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a = an;
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = bn[q];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ys[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * YP + nh * 32 + col] = acc[r];
        }
        __syncthreads();
        // ---- LayerNorm(64) + ReLU on the rows of Y1: 16 lanes per row, one float4 each ----
        {
            const int t16 = tid & 15;
            for (int row = tid >> 4; row < RP; row += 16) {               // all 64 rows: the shuffles need full groups
                const float4 gg = *reinterpret_cast<const float4 *>(lnp + t16 * 4);
                const float4 bb = *reinterpret_cast<const float4 *>(lnp + DD + t16 * 4);
                float4 v = *reinterpret_cast<const float4 *>(ys + row * YP + t16 * 4);
                const float mean = sum16(v.x + v.y + v.z + v.w) * (1.f / DD);
                v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
                const float rstd = rsqrtf(sum16(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.f / DD) + eps);
                *reinterpret_cast<float4 *>(ys + row * YP + t16 * 4) =
                    make_float4(fmaxf(v.x * rstd * gg.x + bb.x, 0.f), fmaxf(v.y * rstd * gg.y + bb.y, 0.f),
                                fmaxf(v.z * rstd * gg.z + bb.z, 0.f), fmaxf(v.w * rstd * gg.w + bb.w, 0.f));
            }
        }
        STORE_K();                            // K2 -> LDS (K1 is no longer read: product 1 ended before the barrier above)
        __syncthreads();
        // the next RoI's X and K1: in flight during product 2 and the last LayerNorm
        if (roi + (int)gridDim.x < K) {
            LOAD_X(roi + gridDim.x);
            LOAD_K(params + (long)(roi + gridDim.x) * p_stride);
        }
        // ---- product 2: Y2[64,256] = Y1[64,64] K2[64,256]; wave = row tile mt, column tiles 4*nh .. 4*nh+3 ----
        {
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            const float *arow = ys + (mt * 32 + col) * YP + half * 4;
            const float *bcol = kw + nh * 128 + col + half * 4 * C;
#pragma unroll 1
            for (int j = 0; j < DD / 8; ++j) {
                const float4 a = *reinterpret_cast<const float4 *>(arow + j * 8);
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bcol[(j * 8 + q) * C + t * 32], acc[t], 0, 0, 0);
            }
            // X was last read in product 1, two barriers ago: xs can take Y2 (rows < R only: the padding stays zero)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < R) xs[row * XP + nh * 128 + t * 32 + col] = acc[t][r];
                }
        }
        __syncthreads();
        // ---- LayerNorm(256) + ReLU on the rows of Y2: 16 lanes per row, float4 number t16 + 16c, coalesced store ----
        {
            const int t16 = tid & 15;
            float *op = out + (long)roi * R * C;
            for (int row = tid >> 4; row < RP; row += 16) {
                const int rr = min(row, R - 1);                           // rows >= R: computed (full shuffle groups), not stored
                float4 v[4];
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = *reinterpret_cast<const float4 *>(xs + rr * XP + (t16 + 16 * c) * 4);
                    s += v[c].x + v[c].y + v[c].z + v[c].w;
                }
                const float mean = sum16(s) * (1.f / C);
                float sq = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean;
                    sq += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
                }
                const float rstd = rsqrtf(sum16(sq) * (1.f / C) + eps);
                if (row < R) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 gg = *reinterpret_cast<const float4 *>(lnp + 2 * DD + (t16 + 16 * c) * 4);
                        const float4 bb = *reinterpret_cast<const float4 *>(lnp + 2 * DD + C + (t16 + 16 * c) * 4);
                        *reinterpret_cast<float4 *>(op + row * C + (t16 + 16 * c) * 4) = make_float4(
                            fmaxf(v[c].x * rstd * gg.x + bb.x, 0.f), fmaxf(v[c].y * rstd * gg.y + bb.y, 0.f),
                            fmaxf(v[c].z * rstd * gg.z + bb.z, 0.f), fmaxf(v[c].w * rstd * gg.w + bb.w, 0.f));
                    }
                }
            }
        }
        __syncthreads();                // xs / kw are rewritten at the top of the loop
    }
}

#undef LOAD_X
#undef LOAD_K
#undef STORE_K

}  // namespace

extern "C" int dfx_dynamic_conv_f32(const float *feats, const float *params, long p_stride, const float *g1,
                                    const float *b1, const float *g2, const float *b2, float *out, int K, int R,
                                    int Cc, int dd, float eps, void *stream)
{
    if (K < 0 || R <= 0) return dfx::fail(DFX_EINVAL, "dynamic_conv: bad dimension");
    if (K == 0) return DFX_OK;
    if (Cc != C || dd != DD || R > RP)
        return dfx::fail(DFX_EINVAL, "dynamic_conv: built for C = 256, dim_dynamic = 64, at most 64 rows per RoI "
                                     "(got C=%d dd=%d R=%d)", Cc, dd, R);
    if (!feats || !params || !g1 || !b1 || !g2 || !b2 || !out) return dfx::fail(DFX_EINVAL, "dynamic_conv: null pointer");
    if (p_stride < 2L * C * DD || (p_stride & 3) || !dfx::aligned16(feats) || !dfx::aligned16(params) ||
        !dfx::aligned16(out) || !dfx::aligned16(g1) || !dfx::aligned16(b1) || !dfx::aligned16(g2) || !dfx::aligned16(b2))
        return dfx::fail(DFX_EINVAL, "dynamic_conv: params rows must hold 2*C*dd floats, 16-byte aligned buffers");
    const size_t lds = sizeof(float) * (RP * XP + C * DD + RP * YP + 2 * DD + 2 * C);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&dynamic_conv<X_F4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&dynamic_conv<X_F4_MAX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return dfx::fail(DFX_ELAUNCH, "dynamic_conv: cannot raise the dynamic LDS limit");
        raised = true;
    }
    const int grid = K < 256 ? K : 256;         // one persistent workgroup per CU
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (R <= 49)
        hipLaunchKernelGGL(dynamic_conv<X_F4>, dim3(grid), dim3(256), lds, st, feats, params, p_stride, g1, b1, g2, b2,
                           out, K, R, eps);
    else
        hipLaunchKernelGGL(dynamic_conv<X_F4_MAX>, dim3(grid), dim3(256), lds, st, feats, params, p_stride, g1, b1, g2,
                           b2, out, K, R, eps);
    return dfx::check_launch("dynamic_conv");
}
