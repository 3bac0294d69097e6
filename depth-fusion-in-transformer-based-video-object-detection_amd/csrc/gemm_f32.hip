// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 256 FLOP/clk/CU) with
// the fused prologue / epilogues of the deformable-attention path (include/dfx_gemm.h).
//
// Structure (CDNA4, wave64):
//   workgroup = 256 threads = 4 waves laid out WM x WN over a BM x BN output tile; a wave owns
//   (BM/WM) x (BN/WN) = MT x NT MFMA tiles of 32 x 32, i.e. MT*NT*16 accumulator registers per lane
//   K is walked in steps of BK = 16 through a double-buffered LDS stage:
//     As[buf][m][k]  - A (and Linear weights, Bs[buf][n][k]) arrive K-contiguous as 16-byte loads and
//                      are stored as they are, one ds_write_b128 each; row pitch BK+4 floats = five
//                      16-byte slots, so the 16 lanes of a ds_read_b128 group (16 different rows, the
//                      same k) fall on 16 different slots
//     Bs[buf][k][n]  - activations of a 1x1 convolution are already [K][N]: 16-byte LDS stores
//   the next K-step's global loads are issued into registers before the 8 x MT*NT MFMAs of the
//   current step, and written to the other LDS buffer after them: one barrier per K-step
//   fragments: the MFMA sums over k in any order as long as A and B agree, so the 8 MFMAs of a
//   K-step are numbered q = 4j + t and lane l (row/col l&31, half h = l>>5) feeds MFMA q with
//   k = 8j + 4h + t: one ds_read_b128 at [row][8j + 4h] yields the lane's operand of FOUR MFMAs
//   (the [K][N] operand reads the same k with ds_read_b32, as its rows run along n)
//   epilogue straight from the accumulators (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31):
//   + bias (per row for convolutions, per column for Linear), + residual, ReLU, zeroing of masked
//   rows; every store instruction writes 2 rows x 128 contiguous bytes.
// MFMA-bound: 2*M*N*K flops against 157 TFLOP/s (fp32 matrix peak, MI355X_MICROARCH.md).
#include "dfx_common.h"
#include "dfx_gemm.h"
#include <stdlib.h>
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Args {
    const float *A, *A2;
    long lda, strideA;
    const float *B;
    long ldb, strideB;
    const float *bias;
    int bias_per_row;
    const float *R;
    long ldr, strideR;
    const unsigned char *mask;
    long strideMask;
    float *C;
    long ldc, strideC;
    int M, N, K, relu;
    int cblk;             // > 0: C is stored column-block-major, [ceil(N / cblk)][M][cblk] (include/dfx_gemm.h)
    long cblk_stride;     // elements between column blocks (>= M * cblk)
    long ablk_stride;     // > 0: A is K-block-major, [K / 4][M][4] with this many elements between blocks
    int wide_epilogue;    // row-major C (and R) with 16-byte aligned rows, N % 4 == 0: float4 epilogue through LDS
    int splits, kper;     // split-K: z = batch * splits + split, split s covers k in [s * kper, min(K, (s + 1) * kper))
    int nx, ny;           // tiles along N and M (the grid is one-dimensional: nx * ny * nz workgroups)
    const float *B2;      // two-segment [K,N] operand: rows k >= K1 come from B2 (row k - K1), same ldb; LDS-DMA path only
    long strideB2;
    int K1;
    const float *ln_g, *ln_b;   // LayerNorm over the N columns after the epilogue's bias / activation / residual (N == BN == 256)
    float ln_eps;
    int act_first;        // the activation applies before the residual is added (y = LN(R + act(A x B + bias)))
    int group_m;          // tile order (placement only, never results): 0 = n fastest, then m, then z, as dispatched;
                          // > 0: workgroup ids are XCD-remapped (each XCD walks one contiguous range) and run m fastest inside
                          // groups of group_m tile rows, then along the columns of every batch element
    int fast_cblk;        // column-block-major C through the same LDS round trip (set by launch())
    int fast_epi;         // wide epilogue in its lean form (set by launch(): no LayerNorm / GELU / act_first, C and R slices < 2 GiB)
    long strideBias;      // elements the column bias advances per batch element (wide column blocks run as a batch: dfx_gemm_f32)
};

__device__ __forceinline__ float activate(float v, int act)      // 1: ReLU, 2: exact (erf) GELU
{
    return act == 1 ? fmaxf(v, 0.f) : 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
}

// waves per SIMD the register allocation must leave room for (HIP's second __launch_bounds__ argument; what the K loop's own
// needs allow: 168 / 120 / 96 / 64 registers per lane for the 128 x 128 / 64 x 128 / 128 x 64 / 64 x 64 tiles, two 8-wave
// workgroups of the 256 x 128 tile; without the bound the scheduler hoists the epilogue's loads over everything and takes
// 2-3x the registers, i.e. a third of the resident waves)
constexpr int min_blocks(int BM, int BN, int NW, int BK)
{
    return BK != 16 ? 1 : NW == 8 ? 4 : BM == 64 && BN == 256 ? 3 : BM == 128 && BN == 128 ? 3 : BM == 64 && BN == 128 ? 4
           : BM == 128 && BN == 64 ? 5 : BM == 64 && BN == 64 ? 6 : BM == 128 && BN == 96 ? 3 : BM == 128 && BN == 32 ? 6 : 1;
}

template <int BM, int BN, int WM, int WN, bool B_KN, int BK, bool DMA>
__global__ __launch_bounds__(64 * WM * WN, min_blocks(BM, BN, WM * WN, BK)) void gemm_f32_kernel(const Args g)
{
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 32, NT = TN / 32;
    constexpr int LDK = BK + 4;                       // [m][k] / [n][k] pitch: 5 (BK = 16) sixteen-byte slots
    constexpr int LDB = BN + 4;                       // [k][n] pitch of the [K][N] operand (16-byte aligned rows)
    constexpr int KQ = BK / 4;                        // float4 per row per K-step
    constexpr int A_F4 = BM * KQ, B_F4 = BN * KQ;     // (= BK * BN / 4 for the [K][N] operand as well)     // float4 per K-step in the A / B tile
    constexpr int NW = WM * WN, NTHR = 64 * NW;       // waves / threads of the workgroup (4 / 256, or 8 / 512 for the 256 x 128 tile)
    constexpr int A_LOADS = (A_F4 + NTHR - 1) / NTHR; // ... per thread (last pass may be partial)
    constexpr int B_LOADS = (B_F4 + NTHR - 1) / NTHR;
    static_assert((NW == 4 || NW == 8) && TM % 32 == 0 && TN % 32 == 0, "bad wave layout");
    static_assert(BK % 8 == 0, "a ds_read_b128 covers 8 consecutive k (4 per lane half)");
    // one LDS object: the two operand stages, re-used by the epilogue as a [64][BN + 4] transpose buffer
    constexpr int A_SZ = BM * LDK, B_SZ = B_KN ? BK * LDB : BN * LDK;               // floats per stage
    constexpr int PR = BN >= 256 ? 32 : 64;           // tile rows per epilogue pass through LDS
    constexpr int LDC = BN + 4, C_SZ = PR * LDC;
    constexpr int S_SZ = 2 * (A_SZ + B_SZ) > C_SZ ? 2 * (A_SZ + B_SZ) : C_SZ;
    __shared__ __attribute__((aligned(16))) float smem[S_SZ];
    float (*const As)[BM][LDK] = reinterpret_cast<float (*)[BM][LDK]>(smem);        // As[buf][m][k]
    float (*const Bs)[B_SZ] = reinterpret_cast<float (*)[B_SZ]>(smem + 2 * A_SZ);   // Bs[buf][...]: [k][n] or [n][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, c = lane & 31;
    // tile of this workgroup (scalar arithmetic; see Args::group_m)
    int bx, by;
    long bz;
    {
        int lin = blockIdx.x;
        if (g.group_m > 0) {
            lin = dfx::xcd_remap(lin, (int)gridDim.x);
            const int nJ = (int)(gridDim.x / (unsigned)g.ny);           // columns of all batch elements
            const int grp = lin / (g.group_m * nJ), y0 = grp * g.group_m;
            const int gsz = min(g.ny - y0, g.group_m), r = lin - grp * g.group_m * nJ;
            by = y0 + r % gsz;
            const int J = r / gsz;
            bx = J % g.nx;
            bz = J / g.nx;
        } else {
            bx = lin % g.nx;
            by = (lin / g.nx) % g.ny;
            bz = lin / (g.nx * g.ny);
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    const long cz = bz;                                   // C slice: one per (batch, split)
    int kbeg = 0, K = g.K;                                // this workgroup's K range
    if (g.splits > 1) {
        const int split = (int)(bz % g.splits);
        bz /= g.splits;
        kbeg = split * g.kper;
        K = min(g.K - kbeg, g.kper);
    }
    const long ashift = g.ablk_stride > 0 ? (long)(kbeg >> 2) * g.ablk_stride : (long)kbeg;
    const float *A = g.A + bz * g.strideA + ashift;
    const float *A2 = g.A2 ? g.A2 + bz * g.strideA + ashift : nullptr;
    const float *B = g.B + bz * g.strideB + (B_KN ? (long)kbeg * g.ldb : (long)kbeg);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Operand staging by buffer loads: one wave-uniform descriptor per operand whose size is the operand's exact
    // extent, a 32-bit byte offset per lane that advances by a constant per K-step.  Rows / columns outside the
    // problem get an offset beyond every extent (0x80000000; the host checks that an operand stays below 2 GiB),
    // so the hardware's range check returns zeros for them - no clamps, no selects, no 64-bit address arithmetic in
    // the loop.  K tails: rows of a [K][N] operand and blocks of a K-block-major A beyond K lie past the extent as
    // well; for K-contiguous operands the last K-step masks its lanes (wave-uniform branch).
    constexpr unsigned kOut = 0x80000000u;
    const long bytesA = g.ablk_stride > 0 ? ((long)(K / 4 - 1) * g.ablk_stride + (long)g.M * 4) * 4
                                          : ((long)(g.M - 1) * g.lda + K) * 4;
    const long bytesB = B_KN ? ((long)(K - 1) * g.ldb + g.N) * 4 : ((long)(g.N - 1) * g.ldb + K) * 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (int)bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A2 ? A2 : A), 0, (int)bytesA, 0x00020000);
    // (two-segment operand: this descriptor covers rows 0 .. K1 - 1, rsB2 the rest)
    const long bytesB1 = (B_KN && g.B2) ? ((long)(g.K1 - 1) * g.ldb + g.N) * 4 : bytesB;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (int)bytesB1, 0x00020000);
    const float *Bsec = (B_KN && g.B2) ? g.B2 + bz * g.strideB2 : B;
    const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(Bsec), 0, (int)(((long)(g.K - g.K1 - 1) * g.ldb + g.N) * 4), 0x00020000);
    unsigned va[A_LOADS], vb[B_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int f = tid + i * NTHR, row = f / KQ, kq = f % KQ, m = m0 + row;
        const unsigned o = g.ablk_stride > 0 ? ((unsigned)kq * (unsigned)g.ablk_stride + (unsigned)m * 4u) * 4u
                                             : ((unsigned)m * (unsigned)g.lda + (unsigned)kq * 4u) * 4u;
        va[i] = (m < g.M && f < A_F4) ? o : kOut;
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int f = tid + i * NTHR;
        if (B_KN) {
            const int kr = f / (BN / 4), nq = f % (BN / 4), n = n0 + nq * 4;
            vb[i] = (n < g.N && f < B_F4) ? ((unsigned)kr * (unsigned)g.ldb + (unsigned)n) * 4u : kOut;
        } else {
            const int row = f / KQ, kq = f % KQ, n = n0 + row;
            vb[i] = (n < g.N && f < B_F4) ? ((unsigned)n * (unsigned)g.ldb + (unsigned)kq * 4u) * 4u : kOut;
        }
    }
    const unsigned stepA = g.ablk_stride > 0 ? (unsigned)(BK / 4) * (unsigned)g.ablk_stride * 4u : BK * 4u;
    const unsigned stepB = B_KN ? (unsigned)BK * (unsigned)g.ldb * 4u : BK * 4u;

    f32x4 ra[A_LOADS], ra2[A_LOADS], rb[B_LOADS];
    bool ktail = false;                   // the loaded K-step reaches past K (wave-uniform)
    int ktail_k0 = 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tiles = [&](int k0) {
        ktail = k0 + BK > K;
        ktail_k0 = k0;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, va[i], 0, 0));
            if (A2) ra2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, va[i], 0, 0));
            va[i] += stepA;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vb[i], 0, 0));
            vb[i] += stepB;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int f = tid + i * NTHR, row = f / KQ, kq = f % KQ;
            if (f >= A_F4) continue;
            f32x4 v = ra[i];
            if (A2) v += ra2[i];
            if (ktail && g.ablk_stride == 0 && ktail_k0 + kq * 4 >= K) v = zero4;
            *reinterpret_cast<f32x4 *>(&As[buf][row][kq * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int f = tid + i * NTHR;
            if (f >= B_F4) continue;
            if (B_KN) {
                const int kr = f / (BN / 4), nq = f % (BN / 4);
                *reinterpret_cast<f32x4 *>(&Bs[buf][kr * LDB + nq * 4]) = rb[i];
            } else {
                const int row = f / KQ, kq = f % KQ;
                f32x4 v = rb[i];
                if (ktail && ktail_k0 + kq * 4 >= K) v = zero4;
                *reinterpret_cast<f32x4 *>(&Bs[buf][row * LDK + kq * 4]) = v;
            }
        }
    };

    // Residual prefetch (64-row tiles only: 8 float4 per thread): a short K (Bottleneck.conv3: 4-16 K-steps) leaves the
    // epilogue's residual loads nothing to hide behind, so they are issued here, before the K loop, through a buffer
    // descriptor of the residual's exact extent (rows / columns outside the problem return zeros, never used).
    constexpr bool PREFETCH_R = BM == 64 && BN == 128;
    constexpr int R_F4 = 64 * BN / 4 / 256;
    f32x4 rpre[PREFETCH_R ? R_F4 : 1];
    const bool use_rpre = PREFETCH_R && g.R && g.wide_epilogue;
    if (PREFETCH_R && use_rpre) {
        const float *Rb = g.R + bz * g.strideR;
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Rb), 0, (int)(((long)(g.M - 1) * g.ldr + g.N) * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < R_F4; ++i) {
            const int f = tid + i * 256, row = f / (BN / 4), c4 = f % (BN / 4);
            const int m = m0 + row, n = n0 + c4 * 4;
            const unsigned o = (m < g.M && n < g.N) ? ((unsigned)m * (unsigned)g.ldr + (unsigned)n) * 4u : 0x80000000u;
            rpre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, o, 0, 0));
        }
    }
    // ---- LDS-DMA staging (DMA = true: no A2 prologue, K a multiple of BK) -------------------------------------------
    // The same LDS images, filled by `buffer_load_dwordx4 ... lds`: the data go from memory straight to LDS, so the
    // VGPR -> LDS store traffic of a K-step (16 KB per workgroup at ~80 B/clk/CU, which delays the other waves' fragment
    // reads: ablation in DESIGN.md, 125 -> 143 TFLOP/s without the stores) disappears, and so do the staging
    // registers.  A wave-instruction fills 64 consecutive 16-byte slots; the images keep their padded pitch (5 slots per
    // K-contiguous row, BN/4 + 1 per [k][n] row), the lane that falls on a pad slot re-loads its neighbour.  Per-lane
    // byte offsets are loop-invariant, the K-step advances through the instruction's scalar offset; rows / columns
    // outside the problem carry an offset beyond the extent and arrive as zeros.
    constexpr int A_SLOTS = BM * (LDK / 4), B_PITCH = B_KN ? LDB / 4 : LDK / 4, B_SLOTS = (B_KN ? BK : BN) * B_PITCH;
    constexpr int A_INSTR = (A_SLOTS + 63) / 64, B_INSTR = (B_SLOTS + 63) / 64;
    constexpr int A_PW = (A_INSTR + NW - 1) / NW, B_PW = (B_INSTR + NW - 1) / NW;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned dva[DMA ? A_PW : 1], dvb[DMA ? B_PW : 1];
    bool dpa[DMA ? A_PW : 1], dpb[DMA ? B_PW : 1];
    if (DMA) {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
            const int sl = (wave_u + NW * i) * 64 + lane, row = sl / (LDK / 4), kq = min(sl % (LDK / 4), BK / 4 - 1), m = m0 + row;
            dpa[i] = wave_u + NW * i < A_INSTR && sl < A_SLOTS;
            const unsigned o = g.ablk_stride > 0 ? ((unsigned)kq * (unsigned)g.ablk_stride + (unsigned)m * 4u) * 4u
                                                 : ((unsigned)m * (unsigned)g.lda + (unsigned)kq * 4u) * 4u;
            dva[i] = m < g.M ? o : kOut;
        }
#pragma unroll
        for (int i = 0; i < B_PW; ++i) {
            const int sl = (wave_u + NW * i) * 64 + lane;
            dpb[i] = wave_u + NW * i < B_INSTR && sl < B_SLOTS;
            if (B_KN) {
                const int kr = sl / B_PITCH, nq = min(sl % B_PITCH, BN / 4 - 1), n = n0 + nq * 4;
                dvb[i] = n < g.N ? ((unsigned)kr * (unsigned)g.ldb + (unsigned)n) * 4u : kOut;
            } else {
                const int row = sl / B_PITCH, kq = min(sl % B_PITCH, BK / 4 - 1), n = n0 + row;
                dvb[i] = n < g.N ? ((unsigned)n * (unsigned)g.ldb + (unsigned)kq * 4u) * 4u : kOut;
            }
        }
    }
    auto dma_tiles = [&](int k0, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass cannot form an LDS-address-space pointer; it only needs the kernel's handle)
        typedef __attribute__((address_space(3))) void *lds_ptr;
        const unsigned sa = g.ablk_stride > 0 ? (unsigned)(k0 >> 2) * (unsigned)g.ablk_stride * 4u : (unsigned)k0 * 4u;
        const bool second = B_KN && g.B2 && k0 >= g.K1;             // (scalar: a K-step lies in one segment, K1 % BK == 0)
        const unsigned sb = B_KN ? (unsigned)(second ? k0 - g.K1 : k0) * (unsigned)g.ldb * 4u : (unsigned)k0 * 4u;
        float *la = &As[buf][0][0], *lb = &Bs[buf][0];
#pragma unroll
        for (int i = 0; i < A_PW; ++i)
            if (dpa[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(la + (wave_u + NW * i) * 256), 16, dva[i], sa, 0, 0);
        if (second) {
#pragma unroll
            for (int i = 0; i < B_PW; ++i)
                if (dpb[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(lb + (wave_u + NW * i) * 256), 16, dvb[i], sb, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < B_PW; ++i)
                if (dpb[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(lb + (wave_u + NW * i) * 256), 16, dvb[i], sb, 0, 0);
        }
#else
        (void)rsB2;
#endif
    };

    const int steps = (K + BK - 1) / BK;
    if (DMA) {
        dma_tiles(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        const int buf = t & 1;
        if (!DMA && t + 1 < steps) load_tiles((t + 1) * BK);   // in flight during the MFMAs below
        // A fragments (and B's for the [N][K] operand): one 16-byte read per lane per 4 MFMAs
        constexpr int KJ = BK / 8;
        float4 af[KJ][MT], bf[KJ][NT];
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[j][i] = *reinterpret_cast<const float4 *>(&As[buf][wm * TM + i * 32 + c][j * 8 + half * 4]);
            if (!B_KN) {
#pragma unroll
                for (int jn = 0; jn < NT; ++jn)
                    bf[j][jn] = *reinterpret_cast<const float4 *>(&Bs[buf][(wn * TN + jn * 32 + c) * LDK + j * 8 + half * 4]);
            }
        }
        float bs[2][NT];                                        // [K][N] operand: scalar reads, one MFMA ahead
        float bsa[DMA && B_KN ? BK / 2 : 1][NT];                // LDS-DMA staging: every fragment of the K-step up front
        if (B_KN && !DMA) {
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) bs[0][jn] = Bs[buf][(half * 4) * LDB + wn * TN + jn * 32 + c];
        }
        if (DMA) {
            // hipcc orders every LDS read that follows an LDS-DMA in program order behind it (s_waitcnt vmcnt(0): it
            // cannot tell the two buffers of the one LDS array apart), so the K-step's reads all come first and the
            // next tile's DMA is issued after them: it then lands under the 8 x MT x NT MFMAs below.
            if (B_KN) {
#pragma unroll
                for (int q = 0; q < BK / 2; ++q)
#pragma unroll
                    for (int jn = 0; jn < NT; ++jn)
                        bsa[q][jn] = Bs[buf][((q >> 2) * 8 + half * 4 + (q & 3)) * LDB + wn * TN + jn * 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < steps) dma_tiles((t + 1) * BK, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
            const int j = q >> 2, tt = q & 3, cur = q & 1, nxt = cur ^ 1;
            if (B_KN && !DMA && q + 1 < BK / 2) {
                const int kn = ((q + 1) >> 2) * 8 + half * 4 + ((q + 1) & 3);
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) bs[nxt][jn] = Bs[buf][kn * LDB + wn * TN + jn * 32 + c];
            }
            // keep the LDS reads of the next MFMA group AHEAD of this group's MFMAs (left alone, the scheduler
            // reuses the fragment registers and sinks the reads below the MFMAs, exposing their latency)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = tt == 0 ? af[j][i].x : tt == 1 ? af[j][i].y : tt == 2 ? af[j][i].z : af[j][i].w;
#pragma unroll
                for (int jn = 0; jn < NT; ++jn) {
                    const float bv = B_KN ? (DMA ? bsa[DMA ? q : 0][jn] : bs[cur][jn])
                                          : (tt == 0 ? bf[j][jn].x : tt == 1 ? bf[j][jn].y : tt == 2 ? bf[j][jn].z : bf[j][jn].w);
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][jn], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!DMA && t + 1 < steps) store_tiles(buf ^ 1);
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next tile have landed ...
        __syncthreads();                                        // ... and after the barrier so have everybody's
    }

    // ---- epilogue ----
#if defined(DFX_GEMM_ABLATE_EPILOGUE)          // timing ablation (tools/r03_exp14.sh): one store per lane keeps the K loop alive
    if (g.M > 0) { g.C[(long)blockIdx.x * NTHR + tid] = acc[0][0][0] + acc[MT - 1][NT - 1][15]; return; }
#endif
    float *C = g.C + cz * g.strideC;
    const float *R = g.R ? g.R + bz * g.strideR : nullptr;
    const unsigned char *mask = g.mask ? g.mask + bz * g.strideMask : nullptr;
    const bool brow = g.bias && g.bias_per_row, bcol = g.bias && !g.bias_per_row;
    const float *const biasp = g.bias ? g.bias + bz * g.strideBias : nullptr;
    constexpr int CQ = BN / 4;                                 // float4 per tile row
    if constexpr (NTHR % CQ == 0) if (g.wide_epilogue && g.fast_epi) {
        // The lean form of the wide epilogue below (same LDS round trip, same float4 rows).  An ablation that ends the tile
        // after the K loop (tools/r03_exp14.sh, profiles/r03_gemm_epilogue_ablation.txt) showed the epilogue costing 13-16 % of
        // a K = 256 launch and 4-5 % of a K = 1024 one - its vector instructions take issue slots from the other resident
        // workgroups' MFMAs - so it is cut to the instructions it needs:
        //   every wave writes one 32-row tile per pass (was: half of the waves two tiles, the others idle);
        //   a thread keeps its column quad (column bias loaded once) and its rows are m = (m0 + r0) + D(pass, it) with D known
        //   at compile time, so a store / residual / row-bias offset is one add to a per-thread base; C, R, the row bias and
        //   the row mask go through buffer descriptors of their exact extents: rows beyond M fall past the extent (loads
        //   return 0, stores are dropped), columns beyond N start from an offset beyond everything - no compares, no selects;
        //   the body is compiled per (bias kind, residual, mask) instead of selecting at run time; ReLU is one v_max each.
        float *Ct = smem;
        constexpr int TPP = PR / (32 * WM);                   // 32-row tiles a wave writes per pass (0: keep the row-range passes)
        constexpr int TP1 = TPP >= 1 ? TPP : 1;
        constexpr bool BAL = TPP >= 1 && PR == TPP * 32 * WM && MT % TP1 == 0;
        constexpr int RS = NTHR / CQ, NIT = PR / RS;          // rows between a thread's float4s, float4s per thread and pass
        static_assert(PR % RS == 0 && (!BAL || (32 * TP1) % RS == 0), "a pass is a whole number of thread rows");
        const int c4 = tid % CQ, r0 = tid / CQ;
        const int n = n0 + c4 * 4, mb = m0 + r0;
        constexpr unsigned kPast = 0x80000000u;
        const bool ncol = n < g.N;
        const unsigned cbase = ncol ? ((unsigned)mb * (unsigned)g.ldc + (unsigned)n) * 4u : kPast;
        const unsigned rbase_off = ncol ? ((unsigned)mb * (unsigned)g.ldr + (unsigned)n) * 4u : kPast;
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)(((long)(g.M - 1) * g.ldc + g.N) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(R ? R : C), 0, (int)(((long)(g.M - 1) * (R ? g.ldr : g.ldc) + g.N) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(biasp ? biasp : C), 0, g.M * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsMask = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(mask ? mask : reinterpret_cast<const unsigned char *>(C)), 0, g.M, 0x00020000);
        const int relu = g.relu;
        auto run = [&](auto bias_kind, auto has_r, auto has_mask) {
            constexpr int BIAS = decltype(bias_kind)::value;          // 0 none, 1 per column, 2 per row
            constexpr bool HAS_R = decltype(has_r)::value, HAS_MASK = decltype(has_mask)::value;
            f32x4 bc = {0.f, 0.f, 0.f, 0.f};
            if (BIAS == 1 && ncol) bc = *reinterpret_cast<const f32x4 *>(biasp + n);
            const bool r_pre = HAS_R && PREFETCH_R && use_rpre;
#pragma unroll
            for (int p = 0; p < BM / PR; ++p) {
                // the residual, row bias and mask of this pass on their way while the tile goes through LDS
                f32x4 rr[HAS_R ? NIT : 1];
                float br[BIAS == 2 ? NIT : 1];
                unsigned char mk[HAS_MASK ? NIT : 1];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int D = BAL ? ((it * RS) / (32 * TP1)) * TM + p * TP1 * 32 + (it * RS) % (32 * TP1) : p * PR + it * RS;
                    if (HAS_R && !r_pre)
                        rr[HAS_R ? it : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, rbase_off + (unsigned)D * (unsigned)g.ldr * 4u, 0, 0));
                    if (BIAS == 2)
                        br[BIAS == 2 ? it : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsBias, (unsigned)(mb + D) * 4u, 0, 0));
                    if (HAS_MASK)
                        mk[HAS_MASK ? it : 0] = __builtin_amdgcn_raw_buffer_load_b8(rsMask, (unsigned)(mb + D), 0, 0);
                }
                if (p > 0) __syncthreads();
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (BAL ? i / TP1 != p : (wm * TM + i * 32) / PR != p) continue;      // compile-time / wave-uniform
                    const int rb = (BAL ? (wm * TPP + i % TP1) * 32 : wm * TM + i * 32 - p * PR) + 4 * half;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            Ct[(rb + (r & 3) + 8 * (r >> 2)) * LDC + wn * TN + j * 32 + c] = acc[i][j][r];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int D = BAL ? ((it * RS) / (32 * TP1)) * TM + p * TP1 * 32 + (it * RS) % (32 * TP1) : p * PR + it * RS;
                    if (m0 + D >= g.M) continue;               // (scalar: the whole thread row lies beyond the last row)
                    f32x4 v = *reinterpret_cast<const f32x4 *>(&Ct[(r0 + it * RS) * LDC + c4 * 4]);
                    if (BIAS == 1) v += bc;
                    if (BIAS == 2) v += br[BIAS == 2 ? it : 0];
                    if (HAS_R) v += r_pre ? rpre[PREFETCH_R ? it : 0] : rr[HAS_R ? it : 0];
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[e]) : "v"(v[e]));
                    }
                    if (HAS_MASK && mk[HAS_MASK ? it : 0]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rsC,
                                                           cbase + (unsigned)D * (unsigned)g.ldc * 4u, 0, 0);
                }
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        using B2 = std::integral_constant<int, 2>;
        auto with_bias = [&](auto has_r, auto has_mask) {
            if (brow) run(B2{}, has_r, has_mask); else if (bcol) run(B1{}, has_r, has_mask); else run(B0{}, has_r, has_mask);
        };
        if (mask) { if (R) with_bias(T{}, T{}); else with_bias(F{}, T{}); }
        else { if (R) with_bias(T{}, F{}); else with_bias(F{}, F{}); }
        return;
    }
    if (g.cblk > 0 && g.fast_cblk) {
        // Column-block-major C ([N / w][M][w], w = 4 or 12: what msda_level_forward reads) through the same LDS round trip: a
        // pass's float4s are numbered block-major - (block, row, quad of the block), quad fastest - so that consecutive lanes
        // write consecutive memory (64 rows x w floats of one block per wave-instruction for w = 4) instead of one scattered
        // dword per lane and accumulator register.
        float *Ct = smem;
        constexpr int TPP = PR / (32 * WM), TP1 = TPP >= 1 ? TPP : 1;
        constexpr bool BAL = TPP >= 1 && PR == TPP * 32 * WM && MT % TP1 == 0;
        constexpr int NF4 = PR * CQ;                           // float4 per pass
        const int QB = g.cblk >> 2, b0 = n0 / g.cblk;         // quads per block, first block of the tile
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)(((long)(g.N / g.cblk - 1) * g.cblk_stride + (long)g.M * g.cblk) * 4), 0x00020000);
        const int relu = g.relu;
#pragma unroll
        for (int p = 0; p < BM / PR; ++p) {
            if (p > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (BAL ? i / TP1 != p : (wm * TM + i * 32) / PR != p) continue;
                const int rb = (BAL ? (wm * TPP + i % TP1) * 32 : wm * TM + i * 32 - p * PR) + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        Ct[(rb + (r & 3) + 8 * (r >> 2)) * LDC + wn * TN + j * 32 + c] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int e0 = 0; e0 < NF4; e0 += NTHR) {
                const int e = e0 + tid;
                if (NF4 % NTHR != 0 && e >= NF4) break;
                const int blk = e / (PR * QB), rem = e - blk * (PR * QB), row = rem / QB, sub = rem - row * QB;
                const int m = BAL ? m0 + (row / (32 * TP1)) * TM + p * TP1 * 32 + row % (32 * TP1) : m0 + p * PR + row;
                const int q = blk * QB + sub, n = n0 + q * 4;
                f32x4 v = *reinterpret_cast<const f32x4 *>(&Ct[row * LDC + q * 4]);
                const bool ok = m < g.M && n < g.N;
                if (bcol) v += *reinterpret_cast<const f32x4 *>(biasp + min(n, g.N - 4));
                if (brow) v += biasp[min(m, g.M - 1)];
                if (relu) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm("v_max_f32 %0, 0, %1" : "=v"(v[u]) : "v"(v[u]));
                }
                if (mask && mask[min(m, g.M - 1)]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                const unsigned o = ok ? (unsigned)(((long)(b0 + blk) * g.cblk_stride + (long)m * g.cblk + sub * 4) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rsC, o, 0, 0);
            }
        }
        return;
    }
    if (g.wide_epilogue) {
        // Row-major C with 16-byte aligned rows: the accumulators (a lane holds one column of 16 scattered rows) go
        // through LDS, 64 tile rows at a time, and leave as float4 per lane - a wave-instruction then covers whole
        // BN*4-byte row segments (512 B for BN = 128) of C and of the residual instead of 128-byte pieces, with a
        // quarter of the memory instructions.  The convolutions with many output channels and a short K
        // (Bottleneck.conv3 + residual) are bound by exactly this traffic.
        float *Ct = smem;
        const bool ln = BN == 256 && g.ln_g != nullptr;       // (a wave-instruction of the float4 loop covers one whole row)
#pragma unroll
        for (int p = 0; p < BM / PR; ++p) {
            if (p > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if ((wm * TM + i * 32) / PR != p) continue;           // wave-uniform
                const int rbase = wm * TM + i * 32 - p * PR + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        Ct[(rbase + (r & 3) + 8 * (r >> 2)) * LDC + wn * TN + j * 32 + c] = acc[i][j][r];
            }
            __syncthreads();
            constexpr int F4 = PR * BN / 4;
#pragma unroll
            for (int f0 = 0; f0 < F4; f0 += NTHR) {
                const int f = f0 + tid;
                if (F4 % NTHR != 0 && f >= F4) break;
                const int row = f / (BN / 4), c4 = f % (BN / 4);
                const int m = m0 + p * PR + row, n = n0 + c4 * 4;
                if (m >= g.M || n >= g.N) continue;
                float4 v = *reinterpret_cast<const float4 *>(&Ct[row * LDC + c4 * 4]);
                if (brow) { const float b = biasp[m]; v.x += b; v.y += b; v.z += b; v.w += b; }
                if (bcol) { const float4 b = *reinterpret_cast<const float4 *>(biasp + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (g.relu && g.act_first) { v.x = activate(v.x, g.relu); v.y = activate(v.y, g.relu); v.z = activate(v.z, g.relu); v.w = activate(v.w, g.relu); }
                if (PREFETCH_R && use_rpre) { const f32x4 q = rpre[PREFETCH_R ? f0 / 256 : 0]; v.x += q[0]; v.y += q[1]; v.z += q[2]; v.w += q[3]; }
                else if (R) { const float4 q = *reinterpret_cast<const float4 *>(R + (long)m * g.ldr + n); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                if (g.relu && !g.act_first) { v.x = activate(v.x, g.relu); v.y = activate(v.y, g.relu); v.z = activate(v.z, g.relu); v.w = activate(v.w, g.relu); }
                if (ln) {
                    // LayerNorm of the row the wave holds (64 lanes x 4 columns): mean, then the centred second moment
                    float sum = (v.x + v.y) + (v.z + v.w);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                    const float mean = sum * (1.f / 256.f);
                    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
                    float sq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
                    const float rstd = 1.f / sqrtf(sq * (1.f / 256.f) + g.ln_eps);
                    const float4 gm = *reinterpret_cast<const float4 *>(g.ln_g + n), bt = *reinterpret_cast<const float4 *>(g.ln_b + n);
                    v.x = v.x * rstd * gm.x + bt.x; v.y = v.y * rstd * gm.y + bt.y;
                    v.z = v.z * rstd * gm.z + bt.z; v.w = v.w * rstd * gm.w + bt.w;
                }
                if (mask && mask[m]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(C + (long)m * g.ldc + n) = v;
            }
        }
        return;
    }
    int ncol[NT];
    float bcolv[NT];
    long coff[NT];                                     // element offset of the column inside a C row
    const long rowmul = g.cblk > 0 ? (long)g.cblk : g.ldc;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        ncol[j] = n0 + wn * TN + j * 32 + c;
        const int nc = min(ncol[j], g.N - 1);
        bcolv[j] = bcol ? biasp[nc] : 0.f;
        coff[j] = g.cblk > 0 ? (long)(nc / g.cblk) * g.cblk_stride + nc % g.cblk : (long)nc;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int mc = min(m, g.M - 1);                         // clamped: branch-free loads
            const float rb = brow ? biasp[mc] : 0.f;
            const bool rz = mask ? mask[mc] != 0 : false;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v = acc[i][j][r] + bcolv[j] + rb;
                if (R) v += R[(long)mc * g.ldr + min(ncol[j], g.N - 1)];
                if (g.relu) v = activate(v, g.relu);
                if (rz) v = 0.f;
                if (m < g.M && ncol[j] < g.N) C[(long)m * rowmul + coff[j]] = v;
            }
        }
    }
}

// Tile order of a launch (Args::group_m): every XCD walks one contiguous range of tiles, 8 tile rows at a time with the
// row fastest, so that the workgroups resident on an XCD share few operand panels in its 4 MiB L2.  Measured at 32 frames
// (tools/bench_gemm.py, DFX_GEMM_GROUP = 0 / 1 / 2 / 4 / 8 / 16 / all rows in one process each, profiles/r03_gemm_order.txt):
// 4 and 8 tie, 2.7 % less time over the path's shapes than the dispatch order (layer4 shortcut 4 %, layer2 conv3 6 %).
int tile_group(const Args &g, int batch)
{
    return dfx::tuning().gemm_group;
}

template <int BM, int BN, int WM, int WN, int BK = 16>
int launch(const Args &g_in, int batch, int b_is_kn, hipStream_t st)
{
    Args g = g_in;
    g.nx = (g.N + BN - 1) / BN;
    g.ny = (g.M + BM - 1) / BM;
    const long total = (long)g.nx * g.ny * batch * (g.splits > 1 ? g.splits : 1);
    if (total >= (1L << 31)) return dfx::fail(DFX_ERANGE, "gemm: too many tiles");
    g.group_m = tile_group(g, batch);
    g.fast_cblk = g.cblk > 0 && g.cblk % 4 == 0 && BN % g.cblk == 0 && g.N % g.cblk == 0 && !g.R && g.relu != 2 && !g.ln_g &&
                  (!g.bias || g.bias_per_row || dfx::aligned16(g.bias)) && dfx::aligned16(g.C) && (g.cblk_stride & 3) == 0 && (g.strideC & 3) == 0 &&
                  ((long)(g.N / g.cblk) * g.cblk_stride) * 4 < (1L << 31) && !dfx::tuning().gemm_old_epilogue;
    // (row offsets of a tile reach up to BM rows past M before the hardware range check drops them: they must not wrap)
    g.fast_epi = g.wide_epilogue && !g.ln_g && !g.act_first && g.relu != 2 && ((long)(g.M + BM) * g.ldc + g.N) * 4 < (1L << 31) &&
                 (!g.R || ((long)(g.M + BM) * g.ldr + g.N) * 4 < (1L << 31)) && !dfx::tuning().gemm_old_epilogue;
    const dim3 grid((unsigned)total), block(64 * WM * WN);
    // measurement aid (dfx_profile_*): flops of the launch in the byte field, tag_a = -1 ([K,N] operand: 1x1 convolution)
    // or -2 (Linear), tag_b = tile
    const long flops = 2L * g.M * g.N * g.K * batch;
    const int kloc = g.splits > 1 ? g.kper : g.K;
    // LDS-DMA staging unless the prologue add needs registers, K has a tail, or the kernel is the HBM-bound short-K
    // residual convolution (layer1 / layer2 conv3), where the DMA wait also drains the residual prefetch
    bool dma = !g.A2 && g.K % BK == 0 && kloc % BK == 0 && !(g.R && kloc <= 128) && !dfx::tuning().gemm_no_dma;
    if (g.B2) {
        if (g.A2 || g.K % BK || g.K1 % BK || g.splits > 1 || !b_is_kn)
            return dfx::fail(DFX_EINVAL, "gemm: a two-segment operand needs K and K1 multiples of %d, no A2, no split-K", BK);
        dma = true;
    }
    if (dma) {
        if (b_is_kn) dfx::launch_timed(flops, -1, BM * 1000 + BN, gemm_f32_kernel<BM, BN, WM, WN, true, BK, true>, grid, block, 0, st, g);
        else dfx::launch_timed(flops, -2, BM * 1000 + BN, gemm_f32_kernel<BM, BN, WM, WN, false, BK, true>, grid, block, 0, st, g);
    } else {
        if (b_is_kn) dfx::launch_timed(flops, -1, BM * 1000 + BN, gemm_f32_kernel<BM, BN, WM, WN, true, BK, false>, grid, block, 0, st, g);
        else dfx::launch_timed(flops, -2, BM * 1000 + BN, gemm_f32_kernel<BM, BN, WM, WN, false, BK, false>, grid, block, 0, st, g);
    }
    return dfx::check_launch("gemm_f32_kernel");
}

// ---- Linears over few rows (the 300-query layers: M = 300 x frames of the rank, N and K a few hundred) ----------------------
// A 64 x 64 tile leaves such a product 76-600 workgroups, each a chain of K / 2 dependent MFMAs per wave behind one memory
// round trip (K = 256: 3.7 us of MFMAs + ~2 us of latency, whatever the tile: profiles/r03_temporal_timeline_F4.txt).  Here a
// workgroup owns ONE 32 x 32 tile of C and its NW waves split K: wave w sums k in [w K/NW, (w+1) K/NW) in chunks of 64, every
// lane loading the 128 contiguous bytes of "its" row of A and of W per chunk straight into the MFMA operand registers (the
// MFMA sums over k in any order as long as A and B agree: lane (c, h) feeds step 4j + t of a chunk with k = 32h + 4j + t) -
// no LDS staging, no K loop barrier, all loads of the wave in flight at once - and the NW partial tiles meet in LDS.
// Same products and the same per-output k order within a wave; the cross-wave sum adds NW partials in wave order.
template <int NW, int CH>
__global__ __launch_bounds__(64 * NW) void linear_rows_kernel(const Args g)
{
    __shared__ __attribute__((aligned(16))) float red[NW][32][36];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int bx = blockIdx.x % g.nx, by = blockIdx.x / g.nx;
    const int m0 = by * 32, n0 = bx * 32;
    constexpr unsigned kOut = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A), 0, (int)(((long)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A2 ? g.A2 : g.A), 0, (int)(((long)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.B), 0, (int)(((long)(g.N - 1) * g.ldb + g.K) * 4), 0x00020000);
    const int kw = wave * (CH * 64) + 32 * h;                      // first k of this lane in chunk 0
    const unsigned oa = m0 + c < g.M ? ((unsigned)(m0 + c) * (unsigned)g.lda + (unsigned)kw) * 4u : kOut;
    const unsigned ob = n0 + c < g.N ? ((unsigned)(n0 + c) * (unsigned)g.ldb + (unsigned)kw) * 4u : kOut;
    f32x4 a[CH][8], b[CH][8];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[ch][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, (ch * 64 + j * 4) * 4, 0));
            b[ch][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ob, (ch * 64 + j * 4) * 4, 0));
        }
    if (g.A2) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                a[ch][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, oa, (ch * 64 + j * 4) * 4, 0));
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ch][j][t], b[ch][j][t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][c] = acc[r];
    __syncthreads();
    if (tid < 256) {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m < g.M && n < g.N) {
            float4 v = *reinterpret_cast<const float4 *>(&red[0][row][c4]);
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float4 q = *reinterpret_cast<const float4 *>(&red[w][row][c4]);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            if (g.wide_epilogue && n + 3 < g.N) {
                if (g.bias) { const float4 q = *reinterpret_cast<const float4 *>(g.bias + n); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                if (g.R) { const float4 q = *reinterpret_cast<const float4 *>(g.R + (long)m * g.ldr + n); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                if (g.relu) { v.x = activate(v.x, g.relu); v.y = activate(v.y, g.relu); v.z = activate(v.z, g.relu); v.w = activate(v.w, g.relu); }
                *reinterpret_cast<float4 *>(g.C + (long)m * g.ldc + n) = v;
            } else {                                           // narrow heads (class / box / reference-point Linears: N = 2 .. 4)
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (n + u >= g.N) break;
                    float x = e[u] + (g.bias ? g.bias[n + u] : 0.f);
                    if (g.R) x += g.R[(long)m * g.ldr + n + u];
                    if (g.relu) x = activate(x, g.relu);
                    g.C[(long)m * g.ldc + n + u] = x;
                }
            }
        }
    }
}

// few rows, [N,K] operand, plain row-major operands and epilogue: K = 64 * NW * CH
bool rows_kernel_applies(const Args &g, int batch, int b_is_kn)
{
    const int max_rows = dfx::tuning().gemm_rows_max;          // tuning aid: the row limit (0 = kernel off)
    const int tiles_n = (g.N + 31) / 32;
    return !b_is_kn && batch == 1 && g.splits <= 1 && !g.mask && !g.cblk && !g.ablk_stride && !g.B2 && !g.ln_g &&
           !g.bias_per_row && (g.M <= max_rows || (max_rows > 0 && g.N <= 128)) && (g.N <= 32 || (g.N % 32 == 0 && g.wide_epilogue)) && g.N <= 1024 &&
           (g.K == 256 || g.K == 512 || g.K == 1024) && (long)((g.M + 31) / 32) * tiles_n <= 9600;
}

int launch_rows(const Args &g_in, hipStream_t st)
{
    Args g = g_in;
    g.nx = (g.N + 31) / 32;
    g.ny = (g.M + 31) / 32;
    const dim3 grid((unsigned)(g.nx * g.ny));
    const long flops = 2L * g.M * g.N * g.K;
    if (g.K == 256) dfx::launch_timed(flops, -2, 32032, linear_rows_kernel<4, 1>, grid, dim3(256), 0, st, g);
    else if (g.K == 512) dfx::launch_timed(flops, -2, 32032, linear_rows_kernel<8, 1>, grid, dim3(512), 0, st, g);
    else dfx::launch_timed(flops, -2, 32032, linear_rows_kernel<8, 2>, grid, dim3(512), 0, st, g);
    return dfx::check_launch("linear_rows_kernel");
}

int choose_and_launch(const Args &g, int batch, int b_is_kn, hipStream_t st);

}  // namespace

extern "C" int dfx_gemm_f32(const float *A, const float *A2, long lda, long strideA, const float *B, long ldb,
                            long strideB, int b_is_kn, const float *bias, int bias_per_row, const float *R,
                            long ldr, long strideR, const unsigned char *row_mask, long strideMask, float *C,
                            long ldc, long strideC, int M, int N, int K, int batch, int relu, int c_block,
                            long c_block_stride, long a_block_stride, void *stream)
{
    if (M < 0 || N < 0 || K <= 0 || batch < 0) return dfx::fail(DFX_EINVAL, "gemm: bad dimension");
    if (relu < 0 || relu > 2) return dfx::fail(DFX_EINVAL, "gemm: activation code must be 0 (none), 1 (ReLU) or 2 (GELU)");
    if ((long)M * N * batch == 0) return DFX_OK;
    if (!A || !B || !C) return dfx::fail(DFX_EINVAL, "gemm: null pointer");
    if ((K & 3) || (lda & 3) || (ldb & 3) || (strideA & 3) || (strideB & 3) || !dfx::aligned16(A) || !dfx::aligned16(B) ||
        (A2 && !dfx::aligned16(A2)) || (b_is_kn && (N & 3)))
        return dfx::fail(DFX_EINVAL, "gemm: K (and N for [K,N] operands) must be multiples of 4, rows 16-byte aligned");
    if (batch > 65535) return dfx::fail(DFX_ERANGE, "gemm: batch too large");
    {   // 32-bit byte offsets inside one batch element's operands (buffer loads)
        const long ea = a_block_stride > 0 ? (long)(K / 4) * a_block_stride : (long)M * lda;
        const long eb = b_is_kn ? (long)K * ldb : (long)N * ldb;
        if (ea * 4 >= (1L << 31) || eb * 4 >= (1L << 31) || (R && (long)M * ldr * 4 >= (1L << 31)))
            return dfx::fail(DFX_ERANGE, "gemm: an operand exceeds 2 GiB per batch element");
    }
    if (c_block < 0 || (c_block > 0 && (c_block_stride < (long)M * c_block || R)))
        return dfx::fail(DFX_EINVAL, "gemm: column-block-major C needs c_block_stride >= M * c_block and no residual");
    if (a_block_stride < 0 || (a_block_stride > 0 && (a_block_stride < (long)M * 4 || (a_block_stride & 3) || A2)))
        return dfx::fail(DFX_EINVAL, "gemm: K-block-major A needs a_block_stride >= 4 * M (a multiple of 4) and no A2");
    if (c_block >= 128 && c_block % 128 == 0 && N % c_block == 0 && N > c_block && batch == 1 && !b_is_kn && !R && a_block_stride == 0 &&
        (c_block_stride & 3) == 0 && dfx::aligned16(C) && (!bias || bias_per_row || dfx::aligned16(bias))) {
        // Wide column blocks ([N / w][M][w] with w a multiple of the tile width: Linears that share their input, stacked along
        // N, each result a contiguous [M, w] tensor of its own - the decoder layers' value projections): every block is a
        // product of its own on the same A.  They run as ONE launch over the batch axis - A fixed, W / the column bias / C
        // advancing per block - in the XCD-aware tile order, which walks the columns of ALL blocks for a group of tile rows
        // before it moves on: an A panel is read once for the whole stack.
        Args g{A, A2, lda, 0, B, ldb, (long)c_block * ldb, bias, bias_per_row, nullptr, 0, 0, row_mask, 0, C, (long)c_block, c_block_stride,
               M, c_block, K, relu, 0, 0, 0, 1, 1, K};
        g.strideBias = bias_per_row ? 0 : c_block;
        return choose_and_launch(g, N / c_block, 0, static_cast<hipStream_t>(stream));
    }
    const int wide = c_block == 0 && (N & 3) == 0 && (ldc & 3) == 0 && (strideC & 3) == 0 && dfx::aligned16(C) &&
                     (!R || ((ldr & 3) == 0 && (strideR & 3) == 0 && dfx::aligned16(R))) &&
                     (!bias || bias_per_row || dfx::aligned16(bias)) && !dfx::tuning().gemm_narrow_epilogue;
    Args g{A, A2, lda, strideA, B, ldb, strideB, bias, bias_per_row, R, ldr, strideR, row_mask, strideMask, C, ldc, strideC,
           M, N, K, relu, c_block, c_block_stride, a_block_stride, wide, 1, K};
    return choose_and_launch(g, batch, b_is_kn, static_cast<hipStream_t>(stream));
}

// C = LayerNorm(R + act(A x B^T + bias)) (or act after the residual) over rows of exactly 256 columns, in ONE launch: the
// Linear that ends a transformer sub-block with its residual add and LayerNorm (include/dfx_gemm.h)
extern "C" int dfx_linear_ln_f32(const float *A, const float *A2, long lda, long a_block_stride, const float *W, long ldw,
                                 const float *bias, const float *R, long ldr, const float *gamma, const float *beta,
                                 float eps, float *C, long ldc, int M, int K, int act, int act_first, void *stream)
{
    const int N = 256;
    if (M < 0 || K <= 0) return dfx::fail(DFX_EINVAL, "linear_ln: bad dimension");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "linear_ln: activation code must be 0, 1 or 2");
    if (M == 0) return DFX_OK;
    if (!A || !W || !C || !gamma || !beta) return dfx::fail(DFX_EINVAL, "linear_ln: null pointer");
    if ((K & 3) || (lda & 3) || (ldw & 3) || (ldc & 3) || (ldr & 3) || !dfx::aligned16(A) || !dfx::aligned16(W) || !dfx::aligned16(C) ||
        (A2 && !dfx::aligned16(A2)) || (R && !dfx::aligned16(R)) || (bias && !dfx::aligned16(bias)) || !dfx::aligned16(gamma) ||
        !dfx::aligned16(beta))
        return dfx::fail(DFX_EINVAL, "linear_ln: K and the leading dimensions must be multiples of 4, buffers 16-byte aligned");
    if (a_block_stride < 0 || (a_block_stride > 0 && (a_block_stride < (long)M * 4 || (a_block_stride & 3) || A2)))
        return dfx::fail(DFX_EINVAL, "linear_ln: K-block-major A needs a_block_stride >= 4 * M (a multiple of 4) and no A2");
    const long ea = a_block_stride > 0 ? (long)(K / 4) * a_block_stride : (long)M * lda;
    if (ea * 4 >= (1L << 31) || (long)N * ldw * 4 >= (1L << 31) || (R && (long)M * ldr * 4 >= (1L << 31)))
        return dfx::fail(DFX_ERANGE, "linear_ln: an operand exceeds 2 GiB");
    Args g{A, A2, lda, 0, W, ldw, 0, bias, 0, R, ldr, 0, nullptr, 0, C, ldc, 0, M, N, K, act, 0, 0, a_block_stride, 1, 1, K};
    g.ln_g = gamma;
    g.ln_b = beta;
    g.ln_eps = eps;
    g.act_first = act_first;
    return launch<64, 256, 1, 4>(g, 1, 0, static_cast<hipStream_t>(stream));
}

// Y[n] = act(W x [X1[n]; X2[n]] + bias): the last 1x1 convolution of a bottleneck and its stride-1 projection shortcut in
// ONE product over the concatenated input channels (include/dfx_gemm.h)
extern "C" int dfx_conv1x1_pair_f32(const float *W, const float *X1, long strideX1, int K1, const float *X2, long strideX2,
                                    int K2, const float *bias, float *Y, long strideY, int Co, int HW, int batch, int act,
                                    void *stream)
{
    if (Co < 0 || HW < 0 || K1 <= 0 || K2 <= 0 || batch < 0) return dfx::fail(DFX_EINVAL, "conv1x1_pair: bad dimension");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "conv1x1_pair: activation code must be 0, 1 or 2");
    if ((long)Co * HW * batch == 0) return DFX_OK;
    if (!W || !X1 || !X2 || !Y) return dfx::fail(DFX_EINVAL, "conv1x1_pair: null pointer");
    const int K = K1 + K2;
    if ((K1 & 15) || (K2 & 15) || (HW & 3) || (strideX1 & 3) || (strideX2 & 3) || (strideY & 3) || !dfx::aligned16(W) ||
        !dfx::aligned16(X1) || !dfx::aligned16(X2) || !dfx::aligned16(Y))
        return dfx::fail(DFX_EINVAL, "conv1x1_pair: channel counts must be multiples of 16, H*W of 4, buffers 16-byte aligned");
    if (batch > 65535 || (long)Co * K * 4 >= (1L << 31) || (long)K1 * HW * 4 >= (1L << 31) || (long)K2 * HW * 4 >= (1L << 31))
        return dfx::fail(DFX_ERANGE, "conv1x1_pair: an operand exceeds 2 GiB per image");
    Args g{W, nullptr, (long)K, 0, X1, (long)HW, strideX1, bias, 1, nullptr, 0, 0, nullptr, 0, Y, (long)HW, strideY,
           Co, HW, K, act, 0, 0, 0, 1, 1, K};
    g.B2 = X2;
    g.strideB2 = strideX2;
    g.K1 = K1;
    return choose_and_launch(g, batch, 1, static_cast<hipStream_t>(stream));
}

namespace {

int choose_and_launch(const Args &g, int batch, int b_is_kn, hipStream_t st)
{
    const int M = g.M, N = g.N, K = g.splits > 1 ? g.kper : g.K;
    const long zb = (long)batch * (g.splits > 1 ? g.splits : 1);
    if (rows_kernel_applies(g, batch, b_is_kn)) return launch_rows(g, st);
    // tile choice.  Small M / N pick the matching narrow tile.
    if (dfx::tuning().gemm_tile >= 0) {      // tuning aid: 0 = 128x128, 1 = 128x64, 2 = 64x128
        const char force[1] = {(char)('0' + dfx::tuning().gemm_tile)};
        if (force[0] == '0') return launch<128, 128, 2, 2>(g, batch, b_is_kn, st);
        if (force[0] == '1') return launch<128, 64, 2, 2>(g, batch, b_is_kn, st);
        if (force[0] == '2') return launch<64, 128, 1, 4>(g, batch, b_is_kn, st);
        if (force[0] == '3') return launch<64, 128, 1, 4, 32>(g, batch, b_is_kn, st);
        if (force[0] == '4') return launch<128, 128, 2, 2, 32>(g, batch, b_is_kn, st);
        if (force[0] == '5') return launch<64, 64, 2, 2>(g, batch, b_is_kn, st);
        if (force[0] == '6') return launch<64, 64, 2, 2, 64>(g, batch, b_is_kn, st);
        if (force[0] == '7') return launch<256, 128, 4, 2>(g, batch, b_is_kn, st);
    }
    if (M <= 64) return launch<64, 128, 1, 4>(g, batch, b_is_kn, st);
    if (N <= 32) return launch<128, 32, 4, 1>(g, batch, b_is_kn, st);
    if (N <= 64) return launch<128, 64, 2, 2>(g, batch, b_is_kn, st);
    if (N <= 96) return launch<128, 96, 4, 1>(g, batch, b_is_kn, st);
    // Measured on the path's shapes at 32 frames (tools/bench_gemm.py with DFX_GEMM_TILE, profiles/r02_bench_gemm_tiles.txt):
    // the 128 x 128 tile (half the operand traffic per flop) is 2-6 % ahead wherever it still gives every CU several
    // tiles and K is deep enough to amortise its longer prologue / epilogue; a short K with a residual epilogue
    // (Bottleneck.conv3 of layer2 / layer3) and M = 64 (layer1) stay on 64 x 128 (6 waves per SIMD).
    {
        const long t128x128 = (long)((M + 127) / 128) * ((N + 127) / 128) * zb;
        const bool rows_fit = (M % 128 == 0) || M >= 1024;
        // split-K launches are sized by the caller to fill one resident round of 128 x 128 tiles (3 per CU)
        if (g.splits > 1 && M >= 128 && N >= 128 && t128x128 <= 768) return launch<128, 128, 2, 2>(g, batch, b_is_kn, st);
        // 8 waves on a 256 x 128 tile (two workgroups = 4 waves per SIMD, a quarter fewer LDS-DMA pieces per MFMA): 2-3 % ahead of
        // 128 x 128 on the 2048-channel convolutions of layer4 and behind it everywhere else (profiles/r03_gemm_tile256.txt:
        // ffn2 +10 %, layer3 +9-11 %, layer4 conv1 +4 %, the short-K shapes far worse)
        if (b_is_kn && M % 256 == 0 && M >= 2048 && K >= 512 && t128x128 >= 8192 && g.splits <= 1)
            return launch<256, 128, 4, 2>(g, batch, b_is_kn, st);
        // the same convolutions on the 4- or 8-frame block of a rank (2112 / 4224 tiles of 128 x 128): 128 x 64 - twice the
        // workgroups, more of them resident - is 14-22 % ahead of 256 x 128 and 5-11 % ahead of 128 x 128 at 4 frames, 4-8 % at 8
        // (tools/r03_exp9.sh, profiles/r03_gemm_tiles_F4_F8.txt)
        if (b_is_kn && M % 128 == 0 && M >= 2048 && K >= 512 && K <= 1536 && t128x128 >= 2048 && g.splits <= 1)
            return launch<128, 64, 2, 2>(g, batch, b_is_kn, st);
        // (since the lean epilogue - residual on its way while the tile crosses LDS - the short-K convolutions with a residual
        // take the 128 x 128 tile too once there are 16 tiles per CU: layer1 / layer2 / layer3 conv3 6.6 / 6.9 / 4.5 % faster at
        // 32 frames, layer1 conv3 10 % at 4 and 8; so do the 256 -> 256 Linears: profiles/r03_gemm_tiles_lean_epilogue.txt)
        if (rows_fit && t128x128 >= 2048 && (K >= 512 || (K >= 256 && !g.R && N >= 256) || (b_is_kn && t128x128 >= 4096)))
            return launch<128, 128, 2, 2>(g, batch, b_is_kn, st);
    }
    // Few tiles (token GEMMs of a 4- or 8-frame rank block): all workgroups are resident at once, the CUs that get
    // one tile more than the others set the time.  A 64 x 64 tile halves that quantum (M = 16800, N = 256:
    // 526 tiles of 64 x 128 = 3 on some CUs, 2.05 on average; 1052 of 64 x 64 = 5 against 4.1).
    auto fill = [](long tiles) { return (double)tiles / (256.0 * (double)((tiles + 255) / 256)); };
    const long t128 = (long)((M + 63) / 64) * ((N + 127) / 128) * zb, t64 = (long)((M + 63) / 64) * ((N + 63) / 64) * zb;
    // At most one 64 x 64 workgroup per CU (the 300-query layers of a small rank block: M = 1200, N = 256): nothing hides
    // the global-load latency of a K-step but the step before it, so the K loop runs at ~1 us per step whatever its
    // depth; 64-deep steps quarter their number (profiles/r02_rank_step.txt).
    if (t64 <= 320 && K >= 128 && !dfx::tuning().gemm_no_deep) return launch<64, 64, 2, 2, 64>(g, batch, b_is_kn, st);
    // deep 1x1 convolutions of an 8-frame block (layer4 conv1 2048 -> 512, layer2 conv1 512 -> 128): 128 x 64 once it gives
    // every CU 8 tiles, 4-5 % ahead of 64 x 64 / 64 x 128 there (profiles/r03_gemm_tiles_F4_F8.txt)
    if (b_is_kn && M % 128 == 0 && K >= 512 && g.splits <= 1 && (long)(M / 128) * ((N + 63) / 64) * zb >= 2048)
        return launch<128, 64, 2, 2>(g, batch, b_is_kn, st);
    // the decoder's first FFN Linear at 32 frames (9600 x 1024 x 256): 1200 tiles of 128 x 64 are 10 % ahead of 64 x 128 / 64 x 64
    // (tools/bench_gemm_queries.py, tools/r03_exp19.sh: 55.6 vs 61.2 / 57.2 us)
    if (!b_is_kn && M % 128 == 0 && N >= 1024 && K <= 256 && g.splits <= 1 && (long)(M / 128) * ((N + 63) / 64) * zb >= 1024)
        return launch<128, 64, 2, 2>(g, batch, b_is_kn, st);
    if (t128 < 8 * 256 && 0.95 * fill(t64) > fill(t128)) return launch<64, 64, 2, 2>(g, batch, b_is_kn, st);
    return launch<64, 128, 1, 4>(g, batch, b_is_kn, st);
}

// ---- split-K ------------------------------------------------------------------------------------------
// out = act(sum_s ws[s] + bias (+ R)), float4 per thread
__global__ void splitk_reduce_kernel(const float *ws, int splits, long MN, int N, const float *bias, int bias_per_row,
                                     const float *R, long ldr, int act, float *C, long ldc)
{
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= MN) return;
    const long e = i4 * 4;
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    float4 v = *reinterpret_cast<const float4 *>(ws + e);
    for (int s = 1; s < splits; ++s) {
        const float4 q = *reinterpret_cast<const float4 *>(ws + (long)s * MN + e);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (bias) {
        if (bias_per_row) { const float b = bias[m]; v.x += b; v.y += b; v.z += b; v.w += b; }
        else { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
    }
    if (R) { const float *r = R + (long)m * ldr + n; v.x += r[0]; v.y += r[1]; v.z += r[2]; v.w += r[3]; }
    if (act) { v.x = activate(v.x, act); v.y = activate(v.y, act); v.z = activate(v.z, act); v.w = activate(v.w, act); }
    float *c = C + (long)m * ldc + n;
    c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
}

}  // namespace

extern "C" int dfx_gemm_splitk_f32(const float *A, long lda, const float *B, long ldb, int b_is_kn, const float *bias,
                                   int bias_per_row, const float *R, long ldr, float *C, long ldc, int M, int N, int K,
                                   int act, int splits, float *workspace, void *stream)
{
    if (M < 0 || N < 0 || K <= 0 || splits < 1) return dfx::fail(DFX_EINVAL, "gemm_splitk: bad dimension");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "gemm_splitk: activation code must be 0 (none), 1 (ReLU) or 2 (GELU)");
    if ((long)M * N == 0) return DFX_OK;
    if (!A || !B || !C || !workspace) return dfx::fail(DFX_EINVAL, "gemm_splitk: null pointer");
    if ((K & 3) || (N & 3) || (lda & 3) || (ldb & 3) || !dfx::aligned16(A) || !dfx::aligned16(B) || !dfx::aligned16(workspace))
        return dfx::fail(DFX_EINVAL, "gemm_splitk: K and N must be multiples of 4, rows and the workspace 16-byte aligned");
    if ((long)M * lda * 4 >= (1L << 31) || (long)(b_is_kn ? K : N) * ldb * 4 >= (1L << 31))
        return dfx::fail(DFX_ERANGE, "gemm_splitk: an operand exceeds 2 GiB");
    int kper = ((K + splits - 1) / splits + 15) / 16 * 16;          // whole K-steps per split
    splits = (K + kper - 1) / kper;
    if (splits > 65535) return dfx::fail(DFX_ERANGE, "gemm_splitk: too many splits");
    Args g{A, nullptr, lda, 0, B, ldb, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, workspace, (long)N, (long)M * N,
           M, N, K, 0, 0, 0, 0, 1, splits, kper};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rc = choose_and_launch(g, 1, b_is_kn, st);
    if (rc != DFX_OK) return rc;
    const long quads = ((long)M * N + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, workspace, splits,
                       (long)M * N, N, bias, bias_per_row, R, ldr, act, C, ldc);
    return dfx::check_launch("splitk_reduce_kernel");
}
