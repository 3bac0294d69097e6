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
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct IgemmArgs {
    const float *X, *Wp;
    const int2 *ktab;          // per k: {tap index ky*KW+kx (or -1: padding column), byte offset ci*H*W*4 + (ky*dil*W + kx*dil)*4}
    const float *bias;
    float *Y;
    int Ci, H, W, Co, Ho, Wo, Kpad, stride, pad, act, KH, KW, dil;
    long strideX, strideY;
    int wide;                  // Ho*Wo % 4 == 0 and y 16-byte aligned: float4 epilogue through LDS
};

__device__ __forceinline__ float activate(float v, int act)
{
    if (act == DFX_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DFX_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}

// UT ("uniform tap"): Ci % 16 == 0, so the 16 k of a K-step are 16 input channels of ONE tap: one table entry, one validity
// test and one offset select per K-step instead of eight of each, the eight loads differ in their scalar offset only.  (The
// fp32 MFMA shares the vector lanes - DESIGN.md section 7 - so the ~50 vector instructions and 7 scalar loads this removes
// per K-step were matrix time.)  Everything but the 7x7 stem and the 1-channel DFormer stem takes this path.
template <int BM, int WM, int WN, bool UT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmArgs g)
{
    constexpr int BN = 128, BK = 16;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int LDK = BK + 4, LDB = BN + 4, LDC = BN + 4;
    constexpr int A_F4 = BM * BK / 4, A_LOADS = (A_F4 + 255) / 256;
    constexpr int B_ROWS = BK / 2;                     // k rows per thread per K-step
    constexpr int A_SZ = BM * LDK, B_SZ = BK * LDB, C_SZ = 64 * LDC;
    constexpr int S_SZ = 2 * (A_SZ + B_SZ) > C_SZ ? 2 * (A_SZ + B_SZ) : C_SZ;
    static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "bad wave layout");
    __shared__ __attribute__((aligned(16))) float smem[S_SZ];
    float (*const As)[BM][LDK] = reinterpret_cast<float (*)[BM][LDK]>(smem);
    float (*const Bs)[B_SZ] = reinterpret_cast<float (*)[B_SZ]>(smem + 2 * A_SZ);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, c = lane & 31;
    // Tile order (placement only): one-dimensional grid, workgroup ids XCD-remapped (each XCD walks one contiguous range,
    // dfx_common.h), the output-channel block fastest - the Co / BM workgroups that gather the SAME input pixels run back to
    // back on one XCD, so the gathered operand comes from HBM once and from that XCD's L2 afterwards (round 3 measured 3.3x the
    // one-pass traffic with the co block as grid.y: its workgroups landed on whatever XCD the dispatch order gave them).
    const int P = g.Ho * g.Wo, HW = g.H * g.W;
    const int ny = (g.Co + BM - 1) / BM, nx = (P + BN - 1) / BN;
    const int lin = dfx::xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int by = lin % ny, rest = lin / ny;
    const int m0 = by * BM, n0 = (rest % nx) * BN;
    const long bz = rest / nx;

    // this thread's output pixel, the top-left input pixel of its receptive field, and a bit per tap that lies
    // inside the map (KH*KW <= 64)
    const int pl = tid & (BN - 1);
    const int kb = __builtin_amdgcn_readfirstlane(tid >> 7);        // 0 / 1: wave-uniform
    const int p = n0 + pl;
    const bool pv = p < P;
    const int oy = (pv ? p : 0) / g.Wo, ox = (pv ? p : 0) - oy * g.Wo;
    const int iy0 = oy * g.stride - g.pad, ix0 = ox * g.stride - g.pad;
    unsigned long long taps = 0;
    if (pv) {
        unsigned rows = 0, cols = 0;
        for (int ky = 0; ky < g.KH; ++ky) rows |= (unsigned)((unsigned)(iy0 + ky * g.dil) < (unsigned)g.H) << ky;
        for (int kx = 0; kx < g.KW; ++kx) cols |= (unsigned)((unsigned)(ix0 + kx * g.dil) < (unsigned)g.W) << kx;
        for (int ky = 0; ky < g.KH; ++ky)
            if ((rows >> ky) & 1u) taps |= (unsigned long long)cols << (ky * g.KW);
    }
    // Buffer loads.  The image descriptor starts `bias_el` elements BEFORE the image, so that the per-lane offset of the
    // receptive field's top-left corner (which lies up to pad rows / columns outside the map) is never negative; the
    // tap's own offset is the scalar offset of the instruction.  A lane whose tap is outside the map gets an offset
    // beyond the extent and the hardware returns 0 for it: no clamp, no select, one shift + compare per element.
    constexpr unsigned kOut = 0x80000000u;
    const int bias_el = g.pad * g.W + g.pad;
    const float *Xn = g.X + bz * g.strideX - bias_el;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Xn), 0, (int)(((long)g.Ci * HW + bias_el) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.Wp), 0, (int)((long)g.Co * g.Kpad * 4), 0x00020000);
    const unsigned vx = (unsigned)(iy0 * g.W + ix0 + bias_el) * 4u;
    unsigned va[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int f = tid + i * 256, row = f / (BK / 4), kq = f % (BK / 4), m = m0 + row;
        va[i] = (m < g.Co && f < A_F4) ? ((unsigned)m * (unsigned)g.Kpad + (unsigned)kq * 4u) * 4u : kOut;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[A_LOADS];
    float rb[B_ROWS];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, va[i], 0, 0));
            va[i] += BK * 4u;
        }
        if (UT) {
            const int2 e = g.ktab[k0 + kb];                              // scalar load: k is wave-uniform
            const unsigned vo = ((taps >> (e.x & 63)) & 1ull) ? vx : kOut;
            const int cstep = 2 * HW * 4;                                // k advances by 2 per row of this thread: 2 channels
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i)
                rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, vo, e.y + i * cstep, 0));
        } else {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const int2 e = g.ktab[k0 + kb + 2 * i];                  // scalar load: k is wave-uniform
                const bool ok = e.x >= 0 && ((taps >> (e.x & 63)) & 1ull);
                rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, ok ? vx : kOut, e.y, 0));
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int f = tid + i * 256, row = f / (BK / 4), kq = f % (BK / 4);
            if (f >= A_F4) continue;
            *reinterpret_cast<f32x4 *>(&As[buf][row][kq * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) Bs[buf][(kb + 2 * i) * LDB + pl] = rb[i];
    };

    const int steps = g.Kpad / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        const int buf = t & 1;
        if (t + 1 < steps) load_tiles((t + 1) * BK);
        constexpr int KJ = BK / 8;
        f32x4 af[KJ][MT];
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[j][i] = *reinterpret_cast<const f32x4 *>(&As[buf][wm * TM + i * 32 + c][j * 8 + half * 4]);
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
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][i][tt], bs[cur][jn], acc[i][jn], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < steps) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float *Y = g.Y + bz * g.strideY;
    if (g.wide && g.act != DFX_ACT_GELU && (long)(g.Co + BM) * P * 4 < (1L << 31)) {
        // The lean epilogue of gemm_f32.hip (see there: the epilogue's vector instructions are matrix time of the other
        // resident workgroups): one 32-row tile per wave and pass, a thread keeps its column quad, rows are m0 + r0 + D(pass, it)
        // with D known at compile time, stores and the row bias through buffer descriptors of the exact extents (rows beyond Co
        // fall past the extent, columns beyond the map start from an offset beyond everything), ReLU = one v_max each.
        float *Ct = smem;
        constexpr int PR = 64, CQ = BN / 4, RS = 256 / CQ, NIT = PR / RS;
        constexpr int TPP = PR / (32 * WM), TP1 = TPP >= 1 ? TPP : 1;
        constexpr bool BAL = TPP >= 1 && PR == TPP * 32 * WM && MT % TP1 == 0;
        static_assert(!BAL || (32 * TP1) % RS == 0, "thread rows must not straddle wave tiles");
        const int c4 = tid % CQ, r0 = tid / CQ, n = n0 + c4 * 4, mb = m0 + r0;
        const unsigned cbase = n < P ? ((unsigned)mb * (unsigned)P + (unsigned)n) * 4u : 0x80000000u;
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((long)g.Co * P * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.bias ? g.bias : Y), 0, g.Co * 4, 0x00020000);
        const bool has_bias = g.bias != nullptr;
        const int relu = g.act == DFX_ACT_RELU;
#pragma unroll
        for (int ps = 0; ps < BM / PR; ++ps) {
            float br[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int D = BAL ? ((it * RS) / (32 * TP1)) * TM + ps * TP1 * 32 + (it * RS) % (32 * TP1) : ps * PR + it * RS;
                br[it] = has_bias ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsBias, (unsigned)(mb + D) * 4u, 0, 0)) : 0.f;
            }
            if (ps > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (BAL ? i / TP1 != ps : (wm * TM + i * 32) / PR != ps) continue;
                const int rb = (BAL ? (wm * TPP + i % TP1) * 32 : wm * TM + i * 32 - ps * PR) + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        Ct[(rb + (r & 3) + 8 * (r >> 2)) * LDC + wn * TN + j * 32 + c] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int D = BAL ? ((it * RS) / (32 * TP1)) * TM + ps * TP1 * 32 + (it * RS) % (32 * TP1) : ps * PR + it * RS;
                if (m0 + D >= g.Co) continue;                  // (scalar: the whole thread row lies beyond the last output channel)
                f32x4 v = *reinterpret_cast<const f32x4 *>(&Ct[(r0 + it * RS) * LDC + c4 * 4]);
                v += br[it];
                if (relu) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm("v_max_f32 %0, 0, %1" : "=v"(v[u]) : "v"(v[u]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rsY,
                                                       cbase + (unsigned)D * (unsigned)P * 4u, 0, 0);
            }
        }
        return;
    }
    if (g.wide) {
        // as gemm_f32.hip: accumulators through LDS, 64 tile rows at a time, out as float4 - whole 512-byte row segments
        float *Ct = smem;
#pragma unroll
        for (int ps = 0; ps < BM / 64; ++ps) {
            if (ps > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if ((wm * TM + i * 32) / 64 != ps) continue;
                const int rbase = wm * TM + i * 32 - ps * 64 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        Ct[(rbase + (r & 3) + 8 * (r >> 2)) * LDC + wn * TN + j * 32 + c] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int f0 = 0; f0 < 64 * BN / 4; f0 += 256) {
                const int f = f0 + tid, row = f / (BN / 4), c4 = f % (BN / 4);
                const int m = m0 + ps * 64 + row, n = n0 + c4 * 4;
                if (m >= g.Co || n >= P) continue;
                float4 v = *reinterpret_cast<const float4 *>(&Ct[row * LDC + c4 * 4]);
                const float bv = g.bias ? g.bias[m] : 0.f;
                v.x = activate(v.x + bv, g.act); v.y = activate(v.y + bv, g.act);
                v.z = activate(v.z + bv, g.act); v.w = activate(v.w + bv, g.act);
                *reinterpret_cast<float4 *>(Y + (long)m * P + n) = v;
            }
        }
        return;
    }
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
    const long blocks = (long)((g.Ho * g.Wo + 127) / 128) * ((g.Co + BM - 1) / BM) * N;
    if (blocks >= (1L << 31)) return dfx::fail(DFX_ERANGE, "conv2d_igemm: too many tiles");
    const dim3 grid((unsigned)blocks), block(256);
    // measurement aid (dfx_profile_*): flops of the launch (K padding included) in the byte field, tag_a = -4
    const long flops = 2L * g.Co * g.Kpad * g.Ho * g.Wo * N;
    if (g.Ci % 16 == 0 && g.Kpad == g.KH * g.KW * g.Ci)
        dfx::launch_timed(flops, -4, BM, conv_igemm_kernel<BM, WM, WN, true>, grid, block, 0, st, g);
    else
        dfx::launch_timed(flops, -4, BM, conv_igemm_kernel<BM, WM, WN, false>, grid, block, 0, st, g);
    return dfx::check_launch("conv_igemm_kernel");
}

}  // namespace

extern "C" int dfx_conv2d_igemm_f32(const float *x, const float *wp, const int *ktab, const float *bias, float *y,
                                    int N, int Ci, int H, int W, int Co, int Ho, int Wo, int Kpad, int KH, int KW,
                                    int stride, int pad, int dilation, int act, long x_image_stride, void *stream)
{
    if (N < 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0 || Ho < 0 || Wo < 0 || Kpad <= 0 || stride <= 0 || pad < 0 ||
        KH <= 0 || KW <= 0 || dilation <= 0)
        return dfx::fail(DFX_EINVAL, "conv2d_igemm: bad dimension");
    if ((long)N * Ho * Wo == 0) return DFX_OK;
    if (!x || !wp || !ktab || !y) return dfx::fail(DFX_EINVAL, "conv2d_igemm: null pointer");
    if (Kpad % 16 || !dfx::aligned16(wp)) return dfx::fail(DFX_EINVAL, "conv2d_igemm: Kpad must be a multiple of 16, wp 16-byte aligned");
    if (KH * KW > 64 || KH > 32 || KW > 32)      // 64-bit tap mask built from 32-bit row and column validity masks
        return dfx::fail(DFX_EINVAL, "conv2d_igemm: at most 64 taps, kernel sides up to 32");
    if (((long)Ci * H * W + (long)pad * W + pad) * 4 >= (1L << 31) || (long)Co * Ho * Wo >= (1L << 31) || (long)Co * Kpad * 4 >= (1L << 31))
        return dfx::fail(DFX_ERANGE, "conv2d_igemm: one image's tensor exceeds 2 GiB");
    if (N > 65535) return dfx::fail(DFX_ERANGE, "conv2d_igemm: batch too large");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "conv2d_igemm: unknown activation");
    const int wide = ((Ho * Wo) & 3) == 0 && dfx::aligned16(y);
    IgemmArgs g{x, wp, reinterpret_cast<const int2 *>(ktab), bias, y, Ci, H, W, Co, Ho, Wo, Kpad, stride, pad, act, KH, KW,
                dilation, x_image_stride > 0 ? x_image_stride : (long)Ci * H * W, (long)Co * Ho * Wo, wide};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Co <= 64) return launch<64, 1, 4>(g, N, st);
    const long t128 = (long)((Co + 127) / 128) * ((Ho * Wo + 127) / 128) * N;
    if (Co % 128 == 0 && t128 >= 2 * 256) return launch<128, 2, 2>(g, N, st);
    return launch<64, 1, 4>(g, N, st);
}
