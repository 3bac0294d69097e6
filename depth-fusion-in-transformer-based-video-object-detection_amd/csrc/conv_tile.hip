// Direct convolution of FEW input channels from an LDS-resident input tile, on the gfx950 matrix cores
// (include/dfx_conv.h, dfx_conv2d_tile_f32): the 7x7/2 ResNet stem (Ci = 3, K = 147) and the first convolution of the DFormer
// depth stem (Ci = 1, 3x3/2, K = 9)  (/root/reference/models/backbone_scratch.py:102-141, dformer_backbone.py:18-71).
//
// The implicit GEMM (conv_igemm.hip) gathers its [K x pixels] operand from global memory K-step by K-step: with few channels
// every k is another TAP of the same few input pixels, so it issues one dword gather per (k, pixel) - 160 per output pixel
// for the stem, 6.2 vector instructions per MFMA, the matrix pipe 0.63 busy (profiles/r03_pmc_gemm.md).  Here the whole
// receptive field of an 8 x 32 output tile ((7 s + KH) x (31 s + KW) input pixels per channel: 21 x 69 x 3 for the stem,
// 17 KB) is staged in LDS ONCE per tile - 9 dword loads per thread - and the gathered operand is read from there:
//
//   workgroup  512 threads = 8 waves, persistent (grid <= 2 per CU); the weights [32 MT][Kpad] are staged in LDS once per
//              workgroup (stem: 64 x 160, 42 KB) and serve every tile it walks
//   tile       8 rows x 32 output pixels x all output channels; wave w owns row w (32 contiguous pixels = the MFMA's N: a
//              store instruction writes whole 128-byte runs of an output row; with 16-pixel rows two workgroups shared
//              every line and the 546 MB output of the depth stem took 448 us instead of 367) and MT 32-channel blocks:
//              MT v_mfma_f32_32x32x2_f32 per k pair, accumulators [co][pixel]
//   B operand  lane (c, h) of MFMA q of a K-step needs x[ci(k), oy s + ky(k), ox s + kx(k)], k = 8j + 4h + t:
//              address = pixel base (per lane, fixed) + tap offset (table in LDS, one int4 per 4 k): one ds_read_b32.
//              Stride 2: the staged rows hold their even columns first, then the odd ones, so that neighbouring output pixels
//              read neighbouring dwords whatever kx is: the 32 lanes of a read group hit 32 consecutive banks.
//   A operand  ds_read_b128 along k of the weight row (pitch Kpad + 4: 16 rows fall on 16 different 16-byte slots)
//   staging    the next tile's input loads are issued before the K loop of the current one and written to the other LDS
//              buffer after it (one barrier per tile); out-of-map pixels are zeros (buffer range check for rows, a select
//              for columns)
//   epilogue   the accumulators start from the bias (staged in LDS); activation straight from them: a lane holds one pixel,
//              16 channels per MFMA tile; the 32 lanes of a row write 128 contiguous bytes; pixel = vector offset,
//              channel = scalar offset of the store: no vector instruction per store
//   tile walk  persistent and wave-uniform: (image, tile row, tile column) advance by scalar adds with carries
// MFMA-bound: 2 * 32 MT * Kpad flops per output pixel against 157 TFLOP/s.
#include "dfx_common.h"
#include "dfx_conv.h"
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kTH = 8, kTW = 32, kThreads = 512, kMaxStage = 9;      // tile: 8 rows x 32 pixels, one row per wave

struct TileArgs {
    const float *X, *Wp, *bias;
    float *Y;
    int Ci, H, W, Co, Ho, Wo, Kpad, K, KH, KW, stride, pad, act;
    long strideX, strideY;
    int TX, TY;            // tiles per output row / column
    long tiles;            // N * TY * TX
    int IH, IW, PW;        // staged input rows / columns per channel, LDS row pitch (dwords)
    int nstage;            // staged elements per thread = ceil(Ci * IH * IW / 512) <= kMaxStage
    unsigned xbytes;       // extent of the input buffer the descriptor covers
};

__device__ __forceinline__ float activate(float v, int act)
{
    if (act == DFX_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DFX_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}

template <int MT, int NS>
__global__ __launch_bounds__(kThreads, 4) void conv_tile_kernel(const TileArgs g)      // 2 workgroups per CU: <= 128 registers
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDA = g.Kpad + 4;
    const int IMG = g.Ci * g.IH * g.PW;                      // dwords of one staged tile
    float *const As = smem;                                   // [32 MT][LDA]
    int *const koff = reinterpret_cast<int *>(smem + 32 * MT * LDA);        // [Kpad]
    float *const img = smem + 32 * MT * LDA + g.Kpad;         // [2][IMG]
    float *const bl = img + 2 * IMG;                          // [32 MT] bias (zeros beyond Co / without a bias)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, c = lane & 31;
    const int s = g.stride, PW = g.PW, PWH = PW >> 1;
    const int HW = g.H * g.W;

    // ---- once per workgroup: weights and the tap table ----
    for (int f = tid; f < 32 * MT * (g.Kpad / 4); f += kThreads) {
        const int row = f / (g.Kpad / 4), kq = f - row * (g.Kpad / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < g.Co) v = *reinterpret_cast<const f32x4 *>(g.Wp + (long)row * g.Kpad + kq * 4);
        *reinterpret_cast<f32x4 *>(As + row * LDA + kq * 4) = v;
    }
    for (int i = tid; i < 32 * MT; i += kThreads) bl[i] = (g.bias && i < g.Co) ? g.bias[i] : 0.f;
    for (int k = tid; k < g.Kpad; k += kThreads) {
        int o = 0;                                            // padded k: zero weight, any valid address
        if (k < g.K) {
            const int tap = k / g.Ci, ci = k - tap * g.Ci, ky = tap / g.KW, kx = tap - ky * g.KW;
            o = ci * g.IH * PW + ky * PW + (s == 2 ? (kx & 1) * PWH + (kx >> 1) : kx);
        }
        koff[k] = o;
    }
    // this lane's pixel of the tile and its base address in the staged image
    const int py = wave, px = c;                               // wave w owns output row w of the tile: 32 contiguous pixels
    const int pixbase = py * s * PW + px;

    // staging plan of this thread: element e = tid + 512 u -> (ci, r, x): LDS index, offset inside the image relative to the
    // tile's top-left input pixel, and the column (for the bounds test) - all independent of the tile
    int s_lds[NS], s_rx[NS];
    const int nel = g.Ci * g.IH * g.IW;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int e = tid + u * kThreads;
        const int ci = e / (g.IH * g.IW), rem = e - ci * (g.IH * g.IW), r = rem / g.IW, x = rem - r * g.IW;
        s_lds[u] = e < nel ? ci * g.IH * PW + r * PW + (s == 2 ? (x & 1) * PWH + (x >> 1) : x) : -1;
        s_rx[u] = (ci << 16) | (r << 8) | x;                  // (IH, IW <= 255: checked on the host)
    }
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.X), 0, (int)g.xbytes, 0x00020000);
    constexpr unsigned kOut = 0xFFFFFFFCu;
    float st[NS];
    // The tile walk is wave-uniform: (image, tile row, tile column) advance by the grid size with two carries - scalar
    // instructions only (a division per tile costs ~40 vector instructions, as many as four K-steps' worth of address work;
    // with 160 MFMAs per wave and tile every per-tile vector instruction counts: the fp32 MFMA shares the lanes)
    const int G = (int)gridDim.x;
    const int d_n = G / (g.TY * g.TX), d_ty = (G - d_n * (g.TY * g.TX)) / g.TX, d_tx = G - d_n * (g.TY * g.TX) - d_ty * g.TX;
    struct Pos { int n, ty, tx; };
    auto advance = [&](Pos p) {
        p.tx += d_tx;
        if (p.tx >= g.TX) { p.tx -= g.TX; p.ty += 1; }
        p.ty += d_ty;
        if (p.ty >= g.TY) { p.ty -= g.TY; p.n += 1; }
        p.n += d_n;
        return p;
    };
    auto stage_load = [&](const Pos &p) {
        const int iy0 = p.ty * kTH * s - g.pad, ix0 = p.tx * kTW * s - g.pad;
        const int base = (int)((long)p.n * g.strideX) + iy0 * g.W + ix0;          // (elements; the batch stays below 2^30 of them)
        // a tile whose staged rectangle lies inside the map needs no bounds test (scalar branch)
        const bool inside = iy0 >= 0 && ix0 >= 0 && iy0 + g.IH <= g.H && ix0 + g.IW <= g.W;
        auto loads = [&](auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                int pk = s_rx[u];
                asm volatile("" : "+v"(pk));                   // (unpacked per tile: hoisted out of the tile loop, the derived values spill)
                const int x = pk & 255, r = (pk >> 8) & 255, ci = pk >> 16;
                unsigned off = (unsigned)(base + (ci * HW + r * g.W + x)) * 4u;
                bool bad = s_lds[u] < 0;                       // (branch-free: selects, no short-circuit)
                if (CHECKED) bad = bad | ((unsigned)(ix0 + x) >= (unsigned)g.W) | ((unsigned)(iy0 + r) >= (unsigned)g.H);
                off = bad ? kOut : off;
                st[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, off, 0, 0));
            }
        };
        if (inside) loads(std::false_type{}); else loads(std::true_type{});
    };
    auto stage_store = [&](int buf) {
        float *dst = img + buf * IMG;
#pragma unroll
        for (int u = 0; u < NS; ++u)
            if (s_lds[u] >= 0) dst[s_lds[u]] = st[u];
    };

    Pos cur;
    {
        const int t0 = (int)blockIdx.x, per = g.TY * g.TX;
        cur.n = t0 / per;
        cur.ty = (t0 - cur.n * per) / g.TX;
        cur.tx = t0 - cur.n * per - cur.ty * g.TX;
    }
    long t = blockIdx.x;
    if (t < g.tiles) {
        stage_load(cur);
        stage_store(0);
    }
    __syncthreads();
    const int nsteps = g.Kpad / 16;
    const unsigned P4 = (unsigned)(g.Ho * g.Wo) * 4u;
    int buf = 0;
    for (; t < g.tiles; t += G) {
        const bool more = t + G < g.tiles;
        const Pos nxt = advance(cur);
        if (more) stage_load(nxt);                           // in flight during the K loop
        f32x16 acc[MT];                                       // start from the bias (staged in LDS once per workgroup)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bl[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        const float *im = img + buf * IMG + pixbase;
        // (two K-steps per iteration on two register sets - the next step's fragments read under the current MFMAs - measured
        // 1.3 % faster on the stem with 12 spilled registers: not kept; four waves per SIMD hide the LDS latency)
#pragma unroll 1
        for (int ks = 0; ks < nsteps; ++ks) {
            const i32x4 o0 = *reinterpret_cast<const i32x4 *>(koff + ks * 16 + half * 4);
            const i32x4 o1 = *reinterpret_cast<const i32x4 *>(koff + ks * 16 + 8 + half * 4);
            f32x4 af[2][MT];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[j][i] = *reinterpret_cast<const f32x4 *>(As + (i * 32 + c) * LDA + ks * 16 + j * 8 + half * 4);
            float bv[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bv[q] = im[o0[q]];
                bv[4 + q] = im[o1[q]];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q >> 2][i][q & 3], bv[q], acc[i], 0, 0, 0);
        }
        // ---- epilogue: activation, one pixel per lane; stores through a descriptor of the image's extent: the lane's pixel
        // is the vector offset (a pixel outside the map starts beyond everything: dropped), the channel the scalar offset
        // (channels beyond Co fall past the extent: dropped) - no vector instruction per store ----
        {
            const int oy = cur.ty * kTH + py, ox = cur.tx * kTW + px;
            const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(g.Y + (long)cur.n * g.strideY, 0, (int)((unsigned)g.Co * P4), 0x00020000);
            const unsigned pix = (oy < g.Ho && ox < g.Wo) ? (unsigned)(oy * g.Wo + ox) * 4u + (unsigned)(4 * half) * P4 : 0x80000000u;
            auto write = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cr = i * 32 + (r & 3) + 8 * (r >> 2);        // + 4 * half: this lane's channel
                        float v = acc[i][r];
                        if (ACT == DFX_ACT_RELU) asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(v));
                        if (ACT == DFX_ACT_GELU) v = activate(v, DFX_ACT_GELU);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, pix, (unsigned)cr * P4, 0);
                    }
            };
            if (g.act == DFX_ACT_RELU) write(std::integral_constant<int, DFX_ACT_RELU>{});
            else if (g.act == DFX_ACT_GELU) write(std::integral_constant<int, DFX_ACT_GELU>{});
            else write(std::integral_constant<int, DFX_ACT_NONE>{});
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        cur = nxt;
    }
}

}  // namespace

extern "C" int dfx_conv2d_tile_fits(int Ci, int Co, int KH, int KW, int stride, int dilation)
{
    if (Ci <= 0 || Co <= 0 || KH <= 0 || KW <= 0) return 0;
    const int K = Ci * KH * KW, Kpad = (K + 15) / 16 * 16;
    const int IH = (kTH - 1) * stride + KH, IW = (kTW - 1) * stride + KW;
    return (stride == 1 || stride == 2) && dilation == 1 && Co <= 64 && Kpad <= 160 && IH <= 255 && IW <= 255 &&
           (Ci * IH * IW + kThreads - 1) / kThreads <= kMaxStage;
}

extern "C" int dfx_conv2d_tile_f32(const float *x, const float *wp, const float *bias, float *y, int N, int Ci, int H, int W,
                                   int Co, int Ho, int Wo, int Kpad, int KH, int KW, int stride, int pad, int act,
                                   long x_image_stride, void *stream)
{
    if (N < 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0 || Ho < 0 || Wo < 0 || Kpad <= 0 || stride <= 0 || pad < 0 || KH <= 0 ||
        KW <= 0)
        return dfx::fail(DFX_EINVAL, "conv2d_tile: bad dimension");
    if ((long)N * Ho * Wo == 0) return DFX_OK;
    if (!x || !wp || !y) return dfx::fail(DFX_EINVAL, "conv2d_tile: null pointer");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "conv2d_tile: unknown activation");
    const int K = Ci * KH * KW;
    if (!dfx_conv2d_tile_fits(Ci, Co, KH, KW, stride, 1) || Kpad != (K + 15) / 16 * 16 || !dfx::aligned16(wp))
        return dfx::fail(DFX_EINVAL, "conv2d_tile: covers stride 1 / 2, Co <= 64, Ci * KH * KW <= 160 (padded to 16), wp 16-byte aligned");
    if ((Ho - 1) * stride - pad + KH > H + pad || (Wo - 1) * stride - pad + KW > W + pad)
        return dfx::fail(DFX_EINVAL, "conv2d_tile: output size does not match the input geometry");
    TileArgs g{};
    g.X = x; g.Wp = wp; g.bias = bias; g.Y = y;
    g.Ci = Ci; g.H = H; g.W = W; g.Co = Co; g.Ho = Ho; g.Wo = Wo; g.Kpad = Kpad; g.K = K; g.KH = KH; g.KW = KW;
    g.stride = stride; g.pad = pad; g.act = act;
    g.strideX = x_image_stride > 0 ? x_image_stride : (long)Ci * H * W;
    g.strideY = (long)Co * Ho * Wo;
    const long xel = (long)(N - 1) * g.strideX + (long)Ci * H * W;
    if (xel * 4 >= (1L << 32) - 8 || (long)Co * Ho * Wo * 4 >= (1L << 32))      // (32-bit byte offsets inside the batch / one output image)
        return dfx::fail(DFX_ERANGE, "conv2d_tile: the input batch or one output image exceeds 4 GiB (split the batch)");
    g.xbytes = (unsigned)(xel * 4);
    g.TY = (Ho + kTH - 1) / kTH;
    g.TX = (Wo + kTW - 1) / kTW;
    g.tiles = (long)N * g.TY * g.TX;
    g.IH = (kTH - 1) * stride + KH;
    g.IW = (kTW - 1) * stride + KW;
    g.PW = stride == 2 ? 2 * ((g.IW + 1) / 2) : g.IW;     // (stride 2: even, the even columns first, then the odd ones)
    g.nstage = (Ci * g.IH * g.IW + kThreads - 1) / kThreads;
    const int MT = Co <= 32 ? 1 : 2;
    const size_t lds = (size_t)(32 * MT * (Kpad + 4) + Kpad + 2 * Ci * g.IH * g.PW + 32 * MT) * 4;
    if (lds > 160 * 1024) return dfx::fail(DFX_EINVAL, "conv2d_tile: the staged tile does not fit the LDS");
    hipStream_t st = static_cast<hipStream_t>(stream);
    static std::mutex mu;
    static bool raised_on[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!raised_on[dev & 63]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_tile_kernel<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_tile_kernel<1, kMaxStage>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_tile_kernel<2, kMaxStage>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return dfx::fail(DFX_ELAUNCH, "conv2d_tile: cannot raise the dynamic LDS limit");
            raised_on[dev & 63] = true;
        }
    }
    static int ncu = 0;
    if (ncu == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        ncu = n;
    }
    // workgroups per CU: what the LDS allows, at most 4 (2048 threads); the one-K-step convolutions (K <= 16) are bound by the
    // latency of a tile's loads and stores, not by the matrix pipe: they want all four
    const long per_cu = lds * 4 <= 160 * 1024 ? 4 : lds * 2 <= 160 * 1024 ? 2 : 1;
    const long grid = g.tiles < ncu * per_cu ? g.tiles : ncu * per_cu;
    // measurement aid (dfx_profile_*): counted as conv_igemm.hip counts the same convolution - 2 * Co * Kpad flops per output pixel
    // (K padding included, channel / tile padding not) -, tag_a = -4 (the direct-convolution family), tag_b = 16 (the tile)
    const long flops = 2L * Co * Kpad * Ho * Wo * N;
    if (MT == 1 && g.nstage <= 3) dfx::launch_timed(flops, -4, 16, conv_tile_kernel<1, 3>, dim3((unsigned)grid), dim3(kThreads), lds, st, g);
    else if (MT == 1) dfx::launch_timed(flops, -4, 16, conv_tile_kernel<1, kMaxStage>, dim3((unsigned)grid), dim3(kThreads), lds, st, g);
    else dfx::launch_timed(flops, -4, 16, conv_tile_kernel<2, kMaxStage>, dim3((unsigned)grid), dim3(kThreads), lds, st, g);
    return dfx::check_launch("conv_tile_kernel");
}
