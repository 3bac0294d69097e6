// Direct convolution as an implicit GEMM on the gfx950 matrix cores (include/dfx_conv.h,
// dfx_conv2d_igemm_f32): the convolutions of the backbones that have no cheaper algorithm here - the 7x7/2
// ResNet stem, the strided 3x3 convolutions, the DFormer depth stem
// (/root/reference/models/backbone_scratch.py:102-141, /root/reference/models/dformer_backbone.py:18-71).
//
//   Y_n[Co, P] = Wp[Co, Kpad] x G_n[Kpad, P],   P = Ho*Wo,   G_n[k, p] = X_n[ci(k), iy(p) + dy(k), ix(p) + dx(k)]
//
// The gathered operand G never exists in memory: a workgroup builds its [BK x BN] slice of it directly in
// LDS.  Thread t owns output pixel n0 + (t & 127) of the tile (so a wave's 64 lanes read 64 neighbouring
// input pixels of one channel / tap: coalesced for stride 1, every other dword for stride 2) and walks 8 of
// the 16 k rows of a K-step; k is wave-uniform, so the tap table entry comes through the scalar cache and
// the bounds test is two unsigned compares per element.  Out-of-map taps read element 0 and are zeroed
// when they are written to LDS, so the loads stay unconditional and in flight across the MFMAs of the
// current K-step.  Everything else is the structure of gemm_f32.hip: 4 waves x (MT x NT) tiles of
// v_mfma_f32_32x32x2_f32, BK = 16 double-buffered, weights [m][k] read by ds_read_b128 along k (one read
// = the operand of four MFMAs), the gathered operand [k][n] by ds_read_b32, epilogue bias + activation
// straight from the accumulators into NCHW.
// MFMA-bound: 2*Co*K*P flops per image against 157 TFLOP/s.
#include "dfx_common.h"
#include "dfx_conv.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct IgemmArgs {
    const float *X, *Wp;
    const int *ktab;
    const float *bias;
    float *Y;
    int Ci, H, W, Co, Ho, Wo, Kpad, stride, pad, act;
    long strideX, strideY;
};

__device__ __forceinline__ float activate(float v, int act)
{
    if (act == DFX_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DFX_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}

template <int BM, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmArgs g)
{
    constexpr int BN = 128, BK = 16;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int LDK = BK + 4, LDB = BN + 4;
    constexpr int A_F4 = BM * BK / 4, A_LOADS = (A_F4 + 255) / 256;
    constexpr int B_ROWS = BK / 2;                     // k rows per thread per K-step
    static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "bad wave layout");
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, c = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long bz = blockIdx.z;
    const float *X = g.X + bz * g.strideX;
    const int P = g.Ho * g.Wo, HW = g.H * g.W;

    // this thread's output pixel and the top-left input pixel of its receptive field
    const int pl = tid & (BN - 1);
    const int kb = __builtin_amdgcn_readfirstlane(tid >> 7);        // 0 / 1: wave-uniform
    const int p = n0 + pl;
    const bool pv = p < P;
    const int oy = (pv ? p : 0) / g.Wo, ox = (pv ? p : 0) - oy * g.Wo;
    const int iy0 = oy * g.stride - g.pad, ix0 = ox * g.stride - g.pad;
    const int pbase = iy0 * g.W + ix0;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_LOADS];
    float rb[B_ROWS];
    unsigned okb = 0;

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int f = tid + i * 256, row = f / (BK / 4), kq = f % (BK / 4);
            const int mc = min(m0 + row, g.Co - 1);                  // rows past Co are never stored
            ra[i] = *reinterpret_cast<const float4 *>(g.Wp + (long)mc * g.Kpad + k0 + kq * 4);
        }
        okb = 0;
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const int e = g.ktab[k0 + kb + 2 * i];                   // scalar load: k is wave-uniform
            const int ci = e & 0xffff, dy = (e >> 16) & 0xff, dx = (e >> 24) & 0x7f;
            const bool ok = pv && e >= 0 && (unsigned)(iy0 + dy) < (unsigned)g.H && (unsigned)(ix0 + dx) < (unsigned)g.W;
            const int off = ci * HW + dy * g.W + dx + pbase;
            rb[i] = X[ok ? off : 0];
            okb |= (unsigned)ok << i;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int f = tid + i * 256, row = f / (BK / 4), kq = f % (BK / 4);
            if (f >= A_F4) continue;
            *reinterpret_cast<float4 *>(&As[buf][row][kq * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) Bs[buf][(kb + 2 * i) * LDB + pl] = (okb >> i) & 1u ? rb[i] : 0.f;
    };

    const int steps = g.Kpad / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        const int buf = t & 1;
        if (t + 1 < steps) load_tiles((t + 1) * BK);
        constexpr int KJ = BK / 8;
        float4 af[KJ][MT];
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[j][i] = *reinterpret_cast<const float4 *>(&As[buf][wm * TM + i * 32 + c][j * 8 + half * 4]);
        float bs[2][NT];
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) bs[0][jn] = Bs[buf][(half * 4) * LDB + wn * TN + jn * 32 + c];
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
            const int j = q >> 2, tt = q & 3, cur = q & 1, nxt = cur ^ 1;
            if (q + 1 < BK / 2) {
                const int kn = ((q + 1) >> 2) * 8 + half * 4 + ((q + 1) & 3);
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) bs[nxt][jn] = Bs[buf][kn * LDB + wn * TN + jn * 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = tt == 0 ? af[j][i].x : tt == 1 ? af[j][i].y : tt == 2 ? af[j][i].z : af[j][i].w;
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bs[cur][jn], acc[i][jn], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < steps) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float *Y = g.Y + bz * g.strideY;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float bv = g.bias ? g.bias[min(m, g.Co - 1)] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * TN + j * 32 + c;
                if (m < g.Co && n < P) Y[(long)m * P + n] = activate(acc[i][j][r] + bv, g.act);
            }
        }
    }
}

template <int BM, int WM, int WN>
int launch(const IgemmArgs &g, int N, hipStream_t st)
{
    const dim3 grid((g.Ho * g.Wo + 127) / 128, (g.Co + BM - 1) / BM, N), block(256);
    // measurement aid (dfx_profile_*): flops of the launch (K padding included) in the byte field, tag_a = -4
    dfx::launch_timed(2L * g.Co * g.Kpad * g.Ho * g.Wo * N, -4, BM, conv_igemm_kernel<BM, WM, WN>, grid, block, 0, st, g);
    return dfx::check_launch("conv_igemm_kernel");
}

}  // namespace

extern "C" int dfx_conv2d_igemm_f32(const float *x, const float *wp, const int *ktab, const float *bias, float *y,
                                    int N, int Ci, int H, int W, int Co, int Ho, int Wo, int Kpad, int stride,
                                    int pad, int act, void *stream)
{
    if (N < 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0 || Ho < 0 || Wo < 0 || Kpad <= 0 || stride <= 0 || pad < 0)
        return dfx::fail(DFX_EINVAL, "conv2d_igemm: bad dimension");
    if ((long)N * Ho * Wo == 0) return DFX_OK;
    if (!x || !wp || !ktab || !y) return dfx::fail(DFX_EINVAL, "conv2d_igemm: null pointer");
    if (Kpad % 16 || !dfx::aligned16(wp)) return dfx::fail(DFX_EINVAL, "conv2d_igemm: Kpad must be a multiple of 16, wp 16-byte aligned");
    if (Ci > 65535 || (long)Ci * H * W >= (1L << 31) || (long)Co * Ho * Wo >= (1L << 31))
        return dfx::fail(DFX_ERANGE, "conv2d_igemm: one image's tensor exceeds 2^31 elements");
    if (N > 65535) return dfx::fail(DFX_ERANGE, "conv2d_igemm: batch too large");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "conv2d_igemm: unknown activation");
    IgemmArgs g{x, wp, ktab, bias, y, Ci, H, W, Co, Ho, Wo, Kpad, stride, pad, act, (long)Ci * H * W, (long)Co * Ho * Wo};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Co <= 64) return launch<64, 1, 4>(g, N, st);
    const long t128 = (long)((Co + 127) / 128) * ((Ho * Wo + 127) / 128) * N;
    if (Co % 128 == 0 && t128 >= 2 * 256) return launch<128, 2, 2>(g, N, st);
    return launch<64, 1, 4>(g, N, st);
}
