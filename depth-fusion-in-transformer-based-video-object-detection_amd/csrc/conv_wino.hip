// 3x3 stride-1 convolution by Winograd F(2x2, 3x3) in ONE kernel on the gfx950 matrix cores
// (include/dfx_conv.h, dfx_conv3x3_wino_f32): the 3x3 convolutions of ResNet-50's bottlenecks, dilated DC5
// stage included (/root/reference/models/backbone_scratch.py:102-141,156-159 -> torchvision Bottleneck.conv2).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input tile d, 3x3 filter g
//
// The element-wise product summed over ci is, for each of the 16 positions of the 4x4 transform domain, a GEMM
//   M_pos[Co, tiles] = U_pos[Co, Ci] x V_pos[Ci, tiles]
// and those 16 GEMMs run on v_mfma_f32_32x32x2_f32 (exact fp32): 2.25x fewer multiplications than the direct
// form, the transforms are additions.  Nothing transformed ever reaches HBM:
//
//   workgroup = 512 threads = 8 waves, output block = 64 output channels x 64 tiles (tiles are numbered
//   through the whole batch, so only the last workgroup has idle lanes); wave w owns transform positions
//   2w and 2w+1 for the whole block: 2 x (2 x 2) MFMA tiles = 128 accumulator registers.
//   K loop over input channels in chunks of 8, double-buffered LDS stage, one barrier per chunk:
//     U chunk  [16 pos][2][64 co][4 ci]  32 KB, contiguous in the pre-blocked weight tensor (16-byte loads,
//              ds_write_b128; a lane's ds_read_b128 = its A operand of the chunk's four MFMAs)
//     V chunk  [16 pos][8 ci][64 tiles]  32 KB: thread (ci = wave, tile = lane) loads its 4x4 input patch
//              straight from the NCHW map (clamped addresses, out-of-map taps zeroed by select), applies
//              B^T d B in registers (32 additions) and stores 16 dwords; B operands by ds_read_b32
//     the next chunk's loads are issued before the current chunk's 32 MFMAs, transformed and stored after.
//   Epilogue: the accumulators of the 8 waves meet in LDS ([16 pos][16 co][64 tiles] per pass, 4 passes,
//   alternating halves of the stage memory), thread (co, tile) gathers its 16 values, applies A^T M A,
//   bias, activation, and writes the 2x2 outputs.
//   Dilation d: the same arithmetic on the d x d interleaved sub-lattices - a tile's outputs are
//   (y0, y0+d) x (x0, x0+d), its taps (y0 + (i-1)d, x0 + (j-1)d).
// MFMA-bound: 2*16*Co*Ci flops per tile against 157 TFLOP/s = 2.25x the direct form's rate at equal speed.
#include "dfx_common.h"
#include "dfx_conv.h"
#include <stdlib.h>

// Diagnostic build only (tools/wino_stamp.py compiles this file with -DDFX_WINO_STAMP into its own library): waves 0 and 4 of
// a few workgroups record s_memtime at fixed points of every K-loop iteration into a buffer no other code reads.
#ifdef DFX_WINO_STAMP
__device__ unsigned long long *g_wino_stamps = nullptr;       // [block slot][wave 0 / 4][chunk][8]
extern "C" int dfx_wino_set_stamp_buffer(void *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define WINO_STAMP(slot)                                                                                   \
    do {                                                                                                   \
        if (stamp_base && lane == 0) stamp_base[ch * 8 + (slot)] = __builtin_amdgcn_s_memtime();           \
    } while (0)
#else
#define WINO_STAMP(slot) do { } while (0)
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct WinoArgs {
    const float *X, *U, *bias;
    float *Y;
    int Ci, H, W, Co, dil, act;
    int TY, TX;          // tiles per image column / row
    long tiles;          // N * TY * TX
    long strideX, strideY;
    int nchunk, ntb;     // Ci / 8, tile blocks
    unsigned xbytes, ubytes;
    int lb0;             // first logical block of this launch
    int no_phase;        // A/B switch: every wave loads before and transforms after its MFMAs (no phase shift between SIMD partners)
};

__device__ __forceinline__ float activate(float v, int act)
{
    if (act == DFX_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DFX_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    return v;
}

constexpr int kCoB = 64, kTB = 64, kCK = 8;       // the weight tensor's blocking: 64 output channels x 8 input channels
constexpr int kUChunk = 16 * 2 * kCoB * 4;        // floats per U chunk of a 64-channel block (8192)

// CB x TB = MFMA tiles (32 output channels x 32 Winograd tiles each) per workgroup and position.  <2,2> is the kernel
// described above; <1,1> - a quarter of the output block, same 8 waves - covers the logical blocks that would
// otherwise form a last, mostly empty round over the 256 CUs (and small problems altogether), sub-block by sub-block.
// (Measured and dropped, round 3: <4,1> = 128 output channels x 32 tiles - half the patch loads / transforms per MFMA, twice
// the LDS-DMA weight traffic, the full 160 KB of LDS - is 6-11 % slower on layer2-4, profiles/r03_wino_wide.txt.)
template <int CB, int TB>
__global__ __launch_bounds__(512) void conv_wino_kernel(const WinoArgs g)
{
    constexpr int CW = 32 * CB, TW = 32 * TB;                 // output channels / tiles of the workgroup
    constexpr int UCH = 16 * 2 * CW * 4, VCH = 16 * kCK * TW; // floats per staged U / V chunk
    constexpr int U_LD = UCH / 4 / 512;                       // float4 loads of U per thread and chunk (4 or 2)
    constexpr int X_ITEMS = kCK * TW;                         // (input channel, tile) patches per chunk (512 or 256)
    constexpr int MS = 16 * 8 * CB * TW;                      // floats of one epilogue pass
    constexpr int SM = 2 * (UCH + VCH) > 2 * MS ? 2 * (UCH + VCH) : 2 * MS;
    __shared__ __attribute__((aligned(16))) float smem[SM];   // 128 KB for <2,2>
    float *const Us = smem;                       // [2][UCH]   [pos][kq][co][4]
    float *const Vs = smem + 2 * UCH;             // [2][VCH]   [pos][ci][tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, c = lane & 31;

    // XCD-aware decode: logical block = co block (slow) x tile block (fast); each XCD walks one contiguous range,
    // so the U slice of a co block stays in that XCD's L2
    int lb, cq = 0, tq = 0;
    if (CB == 2) {
        lb = g.lb0 + dfx::xcd_remap(blockIdx.x, gridDim.x);
    } else {
        const int q = dfx::xcd_remap(blockIdx.x, gridDim.x);
        lb = g.lb0 + (q >> 2);
        cq = (q >> 1) & 1;
        tq = q & 1;
    }
    const int cob = lb / g.ntb, tb = lb - cob * g.ntb;
    const int co0 = cob * kCoB + cq * 32;

    // this thread's tile (the same tile for the input patch it loads and the outputs it writes) and input channel
    const int tl = tid % TW, cil = tid / TW;                  // (tid < X_ITEMS: patch loader)
    const long t = (long)tb * kTB + tq * 32 + tl;
    const bool tv = t < g.tiles;
    const long tc = tv ? t : 0;
    const int per_img = g.TY * g.TX;
    const int n = (int)(tc / per_img);
    const int rem = (int)(tc - (long)n * per_img);
    const int ty = rem / g.TX, tx = rem - ty * g.TX;
    const int d = g.dil;
    // a row's tiles are numbered sub-lattice by sub-lattice (dilation d: d interleaved lattices of step 2d), so that
    // neighbouring lanes are neighbouring tiles OF ONE LATTICE: x0 differs by 2d, and a lane's left / right patch columns are
    // its neighbours' own columns
    const int txl = g.TX / d;
    const int y0 = (ty / d) * 2 * d + ty % d, x0 = (tx % txl) * 2 * d + tx / txl;

    // input patch: clamped row / column offsets + validity bits
    int roff[4], coff[4];
    unsigned okbits = 0;      // bit i: row i valid, bit 4+j: column j valid
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = y0 + (i - 1) * d, xx = x0 + (i - 1) * d;
        if (tv && (unsigned)yy < (unsigned)g.H) okbits |= 1u << i;
        if (tv && (unsigned)xx < (unsigned)g.W) okbits |= 1u << (4 + i);
        roff[i] = min(max(yy, 0), g.H - 1) * g.W;
        coff[i] = min(max(xx, 0), g.W - 1);
    }
    const int HW = g.H * g.W;
    // Patch loads: buffer loads through a wave-uniform descriptor (base, size) with a per-lane 32-bit byte offset and the
    // chunk's scalar offset; no 64-bit address arithmetic.  A thread fetches only the two columns of its patch that no
    // neighbour owns - x0 and x0 + d - and takes column x0 - d from the lane below and x0 + 2d from the lane above
    // (v_mov_b32_dpp wave_shr / wave_shl): every input pixel is requested once per row of tiles instead of twice, in 8 + 4
    // vector memory instructions per chunk instead of 16.  The first and the last lane of a run of tiles fetch their outer
    // column themselves: every lane issues that load, the inner lanes with an offset beyond the buffer (a branch around it
    // makes hipcc wrap the load in a waterfall loop behind an s_waitcnt vmcnt(0)).  Taps outside the map carry the same
    // out-of-range offset, for which the hardware's range check returns 0 without a memory access: no clamp, no select.
    // Why the instruction count matters (tools/wino_stamp.py): the fp32 MFMA runs on the vector lanes, so every cycle a wave
    // spends issuing anything else is lost to the matrix pipe; of its ~3400 cycles per chunk a wave spent 1100-1400 issuing
    // the 16 dword gathers + 4 LDS-DMA pieces of the first version.
    constexpr unsigned kOut = 0xFFFFFFFCu;          // >= the descriptor's extent (the host keeps the input below 2^32 - 4 bytes)
    unsigned xo0[4], xo1[4], xoe[4];
    const bool edge_lo = tl == 0, edge_hi = tl == TW - 1;
    const bool col0 = (okbits >> 4) & 1u, col3 = (okbits >> 7) & 1u;      // x0 - d / x0 + 2d inside the map
    {
        const unsigned b = (unsigned)((long)n * g.strideX) + (unsigned)((cil & (kCK - 1)) * HW);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool row = (okbits >> i) & 1u;
            xo0[i] = (row && ((okbits >> 5) & 1u)) ? (b + (unsigned)(roff[i] + coff[1])) * 4u : kOut;
            xo1[i] = (row && ((okbits >> 6) & 1u)) ? (b + (unsigned)(roff[i] + coff[2])) * 4u : kOut;
            xoe[i] = (row && ((edge_lo && col0) || (edge_hi && col3))) ? (b + (unsigned)(roff[i] + (edge_lo ? coff[0] : coff[3]))) * 4u : kOut;
        }
    }
    const bool xloader = X_ITEMS == 512 || tid < X_ITEMS;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.X), 0, (int)g.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.U), 0, (int)g.ubytes, 0x00020000);
    const unsigned ubase = (unsigned)(co0 / kCoB) * g.nchunk * (kUChunk * 4u);
    // U: the chunk of the 64-channel block is [pos][kq][64 co][4]; this workgroup stages its CW channels of every row
    // (128 channels = two neighbouring 64-channel blocks of the weight tensor)
    unsigned uoff[U_LD];
#pragma unroll
    for (int i = 0; i < U_LD; ++i) {
        const int f = tid + i * 512, row = f / CW, col = cq * 32 + f % CW;
        uoff[i] = (unsigned)(col >> 6) * (unsigned)g.nchunk * (kUChunk * 4u) + (unsigned)((row * kCoB + (col & 63)) * 16);
    }

    f32x16 acc[2][CB][TB];      // [position][co tile][tile tile]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;

    float own0[4], own1[4], oute[4];       // columns x0, x0 + d and (edge lanes) the outer column, rows y0 - d .. y0 + 2d
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // U chunk: memory -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, U_LD per wave): the
    // staged image is lane-linear, so no registers and no ds_write are spent on the weights (half of the chunk's bytes)
    auto dma_u = [&](int ch, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void *lds_ptr;
        const unsigned us = ubase + (unsigned)ch * (kUChunk * 4u);
        float *lu = Us + buf * UCH;
#pragma unroll
        for (int i = 0; i < U_LD; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, (lds_ptr)(lu + (wave_u + 8 * i) * 256), 16, uoff[i], us, 0, 0);
#else
        (void)urs; (void)ubase; (void)wave_u; (void)ch; (void)buf;
#endif
    };
    auto load_x = [&](int ch) {
        const unsigned xs = (unsigned)ch * (unsigned)(kCK * HW) * 4u;             // scalar offset
        if (xloader) {
            // (two dword loads also for d = 1, where the columns are adjacent: hipcc of ROCm 7.2 lowers
            // __builtin_amdgcn_raw_buffer_load_b64 to a single buffer_load_dword - the upper half is never loaded.  The
            // instruction written out as inline asm - buffer_load_dwordx2 into register pairs, an explicit s_waitcnt naming them
            // in store_v - gave correct results and a K loop 15-27 % SLOWER: the compiler brackets every asm load with
            // s_waitcnt for its own pending loads, 30 instead of 11 per iteration, and copies the pairs between the two code
            // paths; round 3, dropped.  One 12 / 16-byte load per row at x0 for both own columns - builtins only, 8 instead of 12
            // vector-memory instructions per chunk - was 8-14 % slower as well: a wide load occupies the address unit in
            // proportion to its bytes per lane, the count of instructions is not what the memory pipe charges)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                own0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo0[i], xs, 0));
                own1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo1[i], xs, 0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) oute[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoe[i], xs, 0));
        }
    };
    auto store_v = [&](int buf) {
        if (!xloader) return;
        // the patch: [left neighbour's x0' + d | own x0 | own x0 + d | right neighbour's x0'']
        // (packed fp32 adds - v_pk_add_f32, two sums per vector instruction - for B^T d B: the rows of the patch are held as
        // column pairs, stage 1 works on whole pairs, stage 2 gets (out0, out3) = pair0 - pair1 in one instruction)
        using f2 = __attribute__((ext_vector_type(2))) float;
        f2 pa[4], pb[4];                                           // row i: (col 0, col 1), (col 2, col 3)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float lo = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own1[i]), 0x138, 0xf, 0xf, true));   // wave_shr:1 (no "old" operand to initialise:
            float hi = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own0[i]), 0x130, 0xf, 0xf, true));   // wave_shl:1  lanes 0 / 63 are edge lanes, overridden below)
            // a neighbour's value stands for an out-of-map column only at the ends of a row of tiles (the lane below / above
            // then belongs to another row): zero there; everything a lane loaded itself is already zero outside the map
            if (edge_lo) lo = oute[i];
            if (edge_hi) hi = oute[i];
            pa[i] = (f2){col0 ? lo : 0.f, own0[i]};
            pb[i] = (f2){own1[i], col3 ? hi : 0.f};
        }
        // B^T d (rows), both column pairs at once
        const f2 ta[4] = {pa[0] - pa[2], pa[1] + pa[2], pa[2] - pa[1], pa[1] - pa[3]};
        const f2 tb[4] = {pb[0] - pb[2], pb[1] + pb[2], pb[2] - pb[1], pb[1] - pb[3]};
        float *vs = Vs + buf * VCH + cil * TW + tl;               // [pos][ci][tile]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f2 o03 = ta[i] - tb[i];                         // (t0 - t2, t1 - t3)
            vs[(i * 4 + 0) * (kCK * TW)] = o03.x;
            vs[(i * 4 + 1) * (kCK * TW)] = ta[i].y + tb[i].x;
            vs[(i * 4 + 2) * (kCK * TW)] = tb[i].x - ta[i].y;    // (as a packed add this pair costs a swizzle and a negation: more instructions)
            vs[(i * 4 + 3) * (kCK * TW)] = o03.y;
        }
    };

    // Two waves share a SIMD (wave w and w + 4).  Their staging work is phase-shifted so that the matrix pipe always has
    // a wave in its MFMA block: waves 0-3 ("early") load chunk t+1 before their MFMAs of chunk t and transform / store it
    // after them; waves 4-7 ("late") transform / store chunk t+1 - loaded one iteration earlier - BEFORE their MFMAs of
    // chunk t, then load chunk t+2.  Either way chunk t+1 is complete at the barrier that ends iteration t, and the
    // buffer it goes to (that of chunk t-1) was released by the previous barrier.
    const bool late = !g.no_phase && __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
#ifdef DFX_WINO_STAMP
    unsigned long long *stamp_base = nullptr;
    if (CB == 2 && g_wino_stamps && (wave == 0 || wave == 4) && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8)
        stamp_base = g_wino_stamps + ((blockIdx.x / 97) * 2 + (wave >> 2)) * (long)(g.nchunk * 8);
#endif
    dma_u(0, 0);
    load_x(0);
    store_v(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (late && g.nchunk > 1) load_x(1);
    __syncthreads();
    for (int ch = 0; ch < g.nchunk; ++ch) {
        const int buf = ch & 1;
        WINO_STAMP(0);
        if (late && ch + 1 < g.nchunk) store_v(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        WINO_STAMP(1);
        // operand fragments of both positions up front (the second position's LDS reads complete under the first
        // position's MFMAs), and BEFORE the next chunk's loads: hipcc orders every LDS access that follows an LDS-DMA in
        // program order behind it (it cannot tell the two buffers apart), so the DMA is issued after the last read
        f32x4 af[2][CB];
        float bf[2][4][TB];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int pos = wave * 2 + p;
            const float *ub = Us + buf * UCH + (pos * 2 + half) * (CW * 4) + c * 4;
            const float *vb = Vs + buf * VCH + (pos * kCK + half * 4) * TW + c;
#pragma unroll
            for (int i = 0; i < CB; ++i) af[p][i] = *reinterpret_cast<const f32x4 *>(ub + i * 32 * 4);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int j = 0; j < TB; ++j) bf[p][tt][j] = vb[tt * TW + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef DFX_WINO_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        WINO_STAMP(2);
        if (ch + 1 < g.nchunk) dma_u(ch + 1, buf ^ 1);           // lands under the MFMAs below
        if (!late) {
            if (ch + 1 < g.nchunk) load_x(ch + 1);
        } else {
            if (ch + 2 < g.nchunk) load_x(ch + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        WINO_STAMP(3);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    const float av = af[p][i][tt];
#pragma unroll
                    for (int j = 0; j < TB; ++j)
                        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[p][tt][j], acc[p][i][j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        WINO_STAMP(4);
        if (!late && ch + 1 < g.nchunk) store_v(buf ^ 1);
        WINO_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the U chunk have landed ...
        WINO_STAMP(6);
        __syncthreads();                            // ... and after the barrier so have everybody's
        WINO_STAMP(7);
    }

    // ---- epilogue: meet in LDS, A^T M A, bias, activation ----
    // pass q takes accumulator registers 4q..4q+3 of every MFMA tile: local co rows i*32 + 8q + 4*half + rr,
    // compacted to cl = i*8 + 4*half + rr
    float *Y = g.Y + (long)n * g.strideY;
    const int P = HW;
    constexpr int ITEMS = 8 * CB * TW;                          // (co row, tile) outputs per pass
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float *ms = smem + (q & 1) * MS;                        // [16 pos][8*CB cl][TW tiles]
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        ms[((wave * 2 + p) * (8 * CB) + i * 8 + 4 * half + rr) * TW + j * 32 + c] = acc[p][i][j][4 * q + rr];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < (ITEMS + 511) / 512; ++s) {
            const int idx = tid + s * 512;
            if (ITEMS < 512 && idx >= ITEMS) break;
            const int cl = idx / TW;                            // (idx % TW == tl: this thread's tile)
            const int co = co0 + (cl >> 3) * 32 + 8 * q + (cl & 7);
            float m[16];
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) m[pos] = ms[(pos * (8 * CB) + cl) * TW + tl];
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
                t1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
            }
            const float bv = g.bias ? g.bias[co] : 0.f;
            float y00 = t0[0] + t0[1] + t0[2] + bv, y01 = t0[1] - t0[2] - t0[3] + bv;
            float y10 = t1[0] + t1[1] + t1[2] + bv, y11 = t1[1] - t1[2] - t1[3] + bv;
            // one scalar branch per item instead of a chain per value (the epilogue's vector instructions are matrix time of
            // the other waves, gemm_f32.hip): ReLU is one v_max each, the GELU polynomial stays out of the way
            if (g.act == DFX_ACT_RELU) {
                asm("v_max_f32 %0, 0, %1" : "=v"(y00) : "v"(y00));
                asm("v_max_f32 %0, 0, %1" : "=v"(y01) : "v"(y01));
                asm("v_max_f32 %0, 0, %1" : "=v"(y10) : "v"(y10));
                asm("v_max_f32 %0, 0, %1" : "=v"(y11) : "v"(y11));
            } else if (g.act == DFX_ACT_GELU) {
                y00 = activate(y00, DFX_ACT_GELU); y01 = activate(y01, DFX_ACT_GELU);
                y10 = activate(y10, DFX_ACT_GELU); y11 = activate(y11, DFX_ACT_GELU);
            }
            if (tv && y0 < g.H && x0 < g.W) {        // (a dilated map's last tile row / column can lie outside it)
                float *yp = Y + (long)co * P + y0 * g.W + x0;
                const bool xv = x0 + d < g.W, yv = y0 + d < g.H;
                yp[0] = y00;
                if (xv) yp[d] = y01;
                if (yv) {
                    yp[d * g.W] = y10;
                    if (xv) yp[d * g.W + d] = y11;
                }
            }
        }
    }
}

// w [Co][Ci][3][3] -> U = G g G^T in the blocked order the kernel stages: [Co/64][Ci/8][16 pos][2][64 co][4 ci]
__global__ void wino_weights_kernel(const float *w, const float *scale, float *u, int Co, int Ci)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Co * Ci) return;
    const int co = idx / Ci, ci = idx - co * Ci;
    const float s = scale ? scale[co] : 1.f;
    float gk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gk[i] = w[(long)idx * 9 + i] * s;
    // G g : 4x3
    float t[12];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t[0 * 3 + j] = gk[0 * 3 + j];
        t[1 * 3 + j] = 0.5f * (gk[0 * 3 + j] + gk[1 * 3 + j] + gk[2 * 3 + j]);
        t[2 * 3 + j] = 0.5f * (gk[0 * 3 + j] - gk[1 * 3 + j] + gk[2 * 3 + j]);
        t[3 * 3 + j] = gk[2 * 3 + j];
    }
    float uu[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uu[i * 4 + 0] = t[i * 3 + 0];
        uu[i * 4 + 1] = 0.5f * (t[i * 3 + 0] + t[i * 3 + 1] + t[i * 3 + 2]);
        uu[i * 4 + 2] = 0.5f * (t[i * 3 + 0] - t[i * 3 + 1] + t[i * 3 + 2]);
        uu[i * 4 + 3] = t[i * 3 + 2];
    }
    const int cob = co / kCoB, col = co % kCoB, chunk = ci / kCK, kq = (ci % kCK) / 4, e = ci % 4;
    const long base = ((long)cob * (Ci / kCK) + chunk) * kUChunk;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) u[base + ((pos * 2 + kq) * kCoB + col) * 4 + e] = uu[pos];
}

}  // namespace

extern "C" int dfx_conv3x3_wino_f32(const float *x, const float *u, const float *bias, float *y, int N, int Ci, int H,
                                    int W, int Co, int dilation, int act, void *stream)
{
    if (N < 0 || Ci <= 0 || H <= 0 || W <= 0 || Co <= 0 || dilation <= 0)
        return dfx::fail(DFX_EINVAL, "conv3x3_wino: bad dimension");
    if (N == 0) return DFX_OK;
    if (!x || !u || !y) return dfx::fail(DFX_EINVAL, "conv3x3_wino: null pointer");
    if (Ci % kCK || Co % kCoB) return dfx::fail(DFX_EINVAL, "conv3x3_wino: Ci must be a multiple of 8, Co of 64");
    if (!dfx::aligned16(u)) return dfx::fail(DFX_EINVAL, "conv3x3_wino: u must be 16-byte aligned");
    if ((long)N * Ci * H * W >= (1L << 30) - 1 || (long)Co * H * W >= (1L << 31))
        return dfx::fail(DFX_ERANGE, "conv3x3_wino: input exceeds 2^30 elements (split the batch)");
    if (act < 0 || act > 2) return dfx::fail(DFX_EINVAL, "conv3x3_wino: unknown activation");
    WinoArgs g{};
    g.X = x; g.U = u; g.bias = bias; g.Y = y;
    g.Ci = Ci; g.H = H; g.W = W; g.Co = Co; g.dil = dilation; g.act = act;
    g.TY = (H + 2 * dilation - 1) / (2 * dilation) * dilation;
    g.TX = (W + 2 * dilation - 1) / (2 * dilation) * dilation;
    g.tiles = (long)N * g.TY * g.TX;
    g.strideX = (long)Ci * H * W;
    g.strideY = (long)Co * H * W;
    g.nchunk = Ci / kCK;
    // the phase shift between SIMD partners pays with many chunks (layer3 / layer4: 1.5 % faster with it, A/B of round 3);
    // with 8 chunks (layer1, Ci = 64) the kernel is 1.4 % faster without it
    // (s_setprio 1 for waves 4-7, the static-priority recipe of the bf16 attention loops: 4-5 % SLOWER here, round 3)
    g.no_phase = dfx::tuning().wino_no_phase >= 0 ? dfx::tuning().wino_no_phase == 1 : g.nchunk <= 8;
    g.xbytes = (unsigned)((long)N * Ci * H * W * 4);
    g.ubytes = (unsigned)((long)16 * Co * Ci * 4);
    const long ntb = (g.tiles + kTB - 1) / kTB;
    const long blocks = ntb * (Co / kCoB);
    if (blocks >= (1L << 29)) return dfx::fail(DFX_ERANGE, "conv3x3_wino: grid too large");
    g.ntb = (int)ntb;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // One 512-thread workgroup per CU: the launch runs in rounds of 256 logical blocks.  A last round that would fill
    // at most ~60 % of the CUs is run as quarter-size workgroups instead (4 per block: ~0.3 of a round when they fit
    // the chip at once); a problem smaller than one round is all quarter-size.
    const long rem = blocks % 256, full = blocks - rem;
    const bool quarter_tail = rem > 0 && rem <= 160 && !dfx::tuning().wino_no_tail;
    const long main_blocks = quarter_tail ? full : blocks;
    const long flops_per_block = 2L * 16 * kCoB * Ci * kTB;
    // measurement aid (dfx_profile_*): MFMA flops the launch executes (16 products per 2x2 output tile, padded tiles
    // included) in the byte field, tag_a = -3, tag_b = dilation; the direct form's flops are 2.25x as many
    if (main_blocks > 0) {
        g.lb0 = 0;
        dfx::launch_timed(flops_per_block * main_blocks, -3, dilation, conv_wino_kernel<2, 2>, dim3((unsigned)main_blocks),
                          dim3(512), 0, st, g);
        const int rc = dfx::check_launch("conv_wino_kernel");
        if (rc != DFX_OK) return rc;
    }
    if (quarter_tail) {
        g.lb0 = (int)full;
        dfx::launch_timed(flops_per_block * rem, -3, dilation, conv_wino_kernel<1, 1>, dim3((unsigned)(rem * 4)), dim3(512), 0,
                          st, g);
    }
    return dfx::check_launch("conv_wino_kernel");
}

extern "C" int dfx_wino_weights_f32(const float *w, const float *scale, float *u, int Co, int Ci, void *stream)
{
    if (Co <= 0 || Ci <= 0 || Ci % kCK || Co % kCoB)
        return dfx::fail(DFX_EINVAL, "wino_weights: Ci must be a multiple of 8, Co of 64");
    if (!w || !u) return dfx::fail(DFX_EINVAL, "wino_weights: null pointer");
    const int total = Co * Ci;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       scale, u, Co, Ci);
    return dfx::check_launch("wino_weights_kernel");
}
