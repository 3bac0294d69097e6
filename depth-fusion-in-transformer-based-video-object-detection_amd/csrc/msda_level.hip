// Fused MSDA with the whole value level resident in LDS (gfx950), for single-level attention.
//
// Geometry: L = 1, M = 8, D = 32, P = 4 - the encoder self-attention, Late Fusion, Encoder Cross
// Fusion and the backbone fusion block of the TransVOD++ RGB-D configuration (one H x W map at
// stride 16: 50 x 84 = 4200 tokens at 800 x 1333).  The wave-per-query kernel (msda_fused.hip)
// fetches every bilinear corner from L2: PMC counters show ~550 MB of L2 requests per 8-frame
// launch for 82 MB of algorithmic traffic, so it runs at the L2 gather rate (~21 TB/s), 0.39 of
// the HBM roofline.  Here the gathers are served by LDS (256 B/clk/CU, ~150 TB/s chip-wide):
//
//   workgroup = (frame, head, channel octet): 8 of the head's 32 channels of EVERY token of the
//               level - (H+3) x (W+2) x 32 B with a zero border, 146 KB of the CU's 160 KB LDS for
//               50 x 84 - staged once with 16-byte loads (each 128-byte value row is read by the
//               4 octet-workgroups of its head, which the block order puts on the same XCD/L2);
//   thread    = one query at a time (1024 threads, queries strided): softmax of the head's 4
//               logits, 4 sampling locations, then 16 corners x 2 ds_read_b128 from LDS and 128
//               FMAs into 8 accumulators; writes its 32 output bytes.
//
// LDS image: two planes of 16-byte chunks, plane c holding channels 4c..4c+3 of every bordered
// token, so the 16 lanes of a ds_read_b128 group - neighbouring queries, neighbouring tokens -
// read 16 consecutive 16-byte slots = all 64 banks once.  The zero border (one token on every
// side) stands for the out-of-map corners of the reference's bilinear rule
// (ms_deform_im2col_cuda.cuh:33-84), so a corner needs no bounds test: the sample-level skip rule
// (-1 < h < H, -1 < w < W, :281-291) zeroes the attention weight instead.  Plane stride PL is
// chosen = 4 mod 8 tokens so that a ds_write_b128 group of the staging pass (4 tokens x 2 planes)
// covers 32 distinct banks.
//
// The tap arithmetic (softmax, location, bilinear weights) is repeated by the 4 octet-workgroups
// of a head; that is the price of fitting the level into LDS, and it overlaps the staging loads of
// the first iteration.  Levels that do not fit (more than ~5100 bordered tokens) and launches
// with few queries per frame (decoder cross-attention, Lq = 300) stay on msda_fused.hip.
#include "dfx_common.h"
#include <stdlib.h>

namespace {

constexpr int STAGE_SLOTS = 10 * 1024;    // 2 chunks x 5120 tokens: the LDS cap (a thread stages STAGE_SLOTS / threads float4)
constexpr long LDS_CAP = 160 * 1024;

typedef float v2f __attribute__((ext_vector_type(2)));      // v_pk_{mul,add,fma}_f32 operands

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }

struct Taps {
    int tb[4];          // y0 * (W+2) + x0 of each point (>= -(W+3)): token index relative to image token (1, 1)
    v2f wt[4], wb[4];   // corner weights {(y0,x0), (y0,x1)} and {(y1,x0), (y1,x1)}, attention weight folded in
};

// The Linear outputs of one (query, head) as loaded: 4 logits, 4 (x, y) offsets, the reference point.
struct Raw {
    float4 lg, o01, o23, r;
};

template <int REFDIM>
__device__ __forceinline__ Raw load_raw(const float *__restrict__ rp, const float *__restrict__ op,
                                        const float *__restrict__ lp)
{
    Raw w;
    w.lg = *reinterpret_cast<const float4 *>(lp);
    w.o01 = *reinterpret_cast<const float4 *>(op);
    w.o23 = *reinterpret_cast<const float4 *>(op + 4);
    if (REFDIM == 2) {
        const float2 r = *reinterpret_cast<const float2 *>(rp);
        w.r = make_float4(r.x, r.y, 0.f, 0.f);
    } else {
        w.r = *reinterpret_cast<const float4 *>(rp);
    }
    return w;
}

// exp(x) for x <= 0 (logit - max), two at a time: 2^t * (1 + f ln2) with t = RN(x log2e) and f the
// rounding error of that product (an fma away) - v_exp_f32's 1 ulp plus ~1e-8 |x|, without libm's
// range reduction (x is clamped at -87, where the result is 1e-38).
__device__ __forceinline__ v2f exp_neg(v2f x)
{
    x.x = fmaxf(x.x, -87.f);
    x.y = fmaxf(x.y, -87.f);
    const v2f L2E = splat(1.4426950216293335f);
    const v2f t = x * L2E;
    const v2f f = pk_fma(x, L2E, -t) * splat(0.6931471805599453f);
    const v2f e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return pk_fma(e, f, e);
}

// Per-thread constants of the level.
struct Level {
    v2f size, rsize;    // (W, H) and their correctly rounded reciprocals
    int H, W, WB;
};

// One (query, head): softmax over the 4 logits (F.softmax, ms_deform_attn.py:99), locations
// ref + off / (W, H) or ref_xy + off / P * ref_wh * 0.5 (:102-110), pixel coordinates and bilinear
// weights (ms_deform_im2col_cuda.cuh:33-84, :281-291).  Written on (x, y) pairs so that it compiles
// to packed fp32 instructions; the two divisions per point are q = o*r, q += (o - q*size)*r with
// r = RN(1/size), which is the correctly rounded quotient (Markstein), and the softmax quotient is
// the same iteration on a Newton-refined v_rcp_f32.
template <int REFDIM>
__device__ __forceinline__ Taps make_taps(const Raw &in, const Level &lv)
{
    const float mx = fmaxf(fmaxf(in.lg.x, in.lg.y), fmaxf(in.lg.z, in.lg.w));
    const v2f e01 = exp_neg((v2f){in.lg.x - mx, in.lg.y - mx});
    const v2f e23 = exp_neg((v2f){in.lg.z - mx, in.lg.w - mx});
    float sum = 0.f;
    sum += e01.x; sum += e01.y; sum += e23.x; sum += e23.y;
    float y0 = __builtin_amdgcn_rcpf(sum);
    y0 = fmaf(fmaf(-sum, y0, 1.f), y0, y0);
    const v2f ys = splat(y0), ss = splat(sum);
    v2f a01 = e01 * ys, a23 = e23 * ys;
    a01 = pk_fma(pk_fma(-a01, ss, e01), ys, a01);
    a23 = pk_fma(pk_fma(-a23, ss, e23), ys, a23);
    const float aw[4] = {a01.x, a01.y, a23.x, a23.y};
    const v2f o[4] = {{in.o01.x, in.o01.y}, {in.o01.z, in.o01.w}, {in.o23.x, in.o23.y}, {in.o23.z, in.o23.w}};
    const v2f rxy = {in.r.x, in.r.y}, rwh = {in.r.z, in.r.w};
    Taps t;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        v2f loc;
        if (REFDIM == 2) {
            v2f q = o[p] * lv.rsize;
            q = pk_fma(pk_fma(-q, lv.size, o[p]), lv.rsize, q);
            loc = rxy + q;
        } else {
            loc = rxy + o[p] * splat(0.25f) * rwh * splat(0.5f);
        }
        const v2f im = pk_fma(loc, lv.size, splat(-0.5f));               // (w_im, h_im)
        // clamp to [-1, size]: NaN and -inf land on -1 (zero border, weight 0), +inf on `size` (dropped below)
        const float ws = __builtin_amdgcn_fmed3f(im.x, -1.f, lv.size.x);
        const float hs = __builtin_amdgcn_fmed3f(im.y, -1.f, lv.size.y);
        int ix, iy;                                                      // floor to int in one instruction
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(ws));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(hs));
        const v2f l = {__builtin_amdgcn_fractf(ws), __builtin_amdgcn_fractf(hs)};   // (lw, lh) = x - floor(x), exact
        const v2f hc = splat(1.f) - l;                                   // (hw, hh)
        // the skip rule -1 < h_im < H, -1 < w_im < W: a sample at exactly -1 has weight 0 on its only
        // in-map row/column already, so only the upper bounds are left to test
        const float aa = (iy < lv.H && ix < lv.W) ? aw[p] : 0.f;
        const v2f xw = {hc.x, l.x};                                      // (hw, lw)
        t.wt[p] = splat(hc.y) * xw * splat(aa);
        t.wb[p] = splat(l.y) * xw * splat(aa);
        // iy * WB + ix; the (+1, +1) of the border is folded into the image base by the caller (rows 0..H+1)
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t.tb[p]) : "v"(iy), "v"(lv.WB), "v"(ix));
    }
    return t;
}

__device__ __forceinline__ void fma4(float4 &a, float w, const float4 &v)
{
    a.x = fmaf(w, v.x, a.x);
    a.y = fmaf(w, v.y, a.y);
    a.z = fmaf(w, v.z, a.z);
    a.w = fmaf(w, v.w, a.w);
}

// Operand strides (include/dfx_msda.h, dfx_msda_level_layout).  The reference layouts (value
// [N,S,8,32], one row of Linear outputs per query, out [N,Lq,256]) work, but a workgroup then uses 32
// of every 128-byte line it pulls from L2 and scatters its stores; the layouts the kernel is built for
// are the block-major ones dfx_gemm_f32 writes / reads (c_block 4 and 12, a_block), where
// everything a workgroup touches is contiguous.
struct LevelArgs {
    const float *value, *ref, *off, *logits;
    float *out;
    dfx_msda_level_layout ly;
    int H, W, Lq, PL, qsplit, nitems;
};

// Persistent: the grid is at most one workgroup per CU (146 KB of LDS each) and every workgroup walks the items
// (frame, head, octet, query slice) b, b + grid, b + 2 grid, ...: workgroup launch, index setup and the zero border are
// paid once per CU instead of once per item (4 items per CU at 32 frames).
// THREADS / PREFETCH (round 4): <1024, false> is the kernel of rounds 1-3 - stage, barrier, gather, barrier, item after item.  Its
// phases add up (tools/level_ablate.py, profiles/r04_level_ablate.txt: staging ~35 us, tap arithmetic + parameters + stores
// ~34 us, gather ~18 us of a 97 us launch at 32 frames), because the 146 KB image leaves no LDS to stage item i + 1 beside the
// gather of item i.  <512, true> stages through REGISTERS instead: a workgroup of 8 waves has 256 registers per thread, 80 of
// them hold the next item's level (20 float4 per thread) - its loads are issued in slices inside the gather loop and written
// to LDS when the gather is done.  Correct (tests/test_msda_gpu.py::test_level_kernel_prefetch_variant) and SLOWER: 125 vs 116 us
// cold, 105 vs 99 us warm at 32 frames - with half the waves per CU the two compute phases lose what the overlap gains.  Kept
// as a selectable variant (DFX_LEVEL_VARIANT=1); the default is <1024, false>.
template <int REFDIM, int THREADS, bool PREFETCH>
__global__ __launch_bounds__(THREADS) void msda_fused_level(const LevelArgs g)
{
    constexpr int STAGE_PASSES = STAGE_SLOTS / THREADS;
    const int H = g.H, W = g.W, Lq = g.Lq, PL = g.PL, qsplit = g.qsplit;
    extern __shared__ float4 img[];                 // [2 planes][PL bordered tokens]
    const int tid = threadIdx.x;
    const int S = H * W, WB = W + 2;
    const int per = 4 * qsplit;
    const int qper = (Lq + qsplit - 1) / qsplit;
    Level lv;
    lv.size = (v2f){(float)W, (float)H};
    lv.rsize = (v2f){1.f / (float)W, 1.f / (float)H};
    lv.H = H; lv.W = W; lv.WB = WB;
    const long vs_token = g.ly.value_token, vs_chunk = g.ly.value_chunk;
    const long off_stride = g.ly.off_row, logit_stride = g.ly.logit_row;
    const long out_row = g.ly.out_row, out_chunk = g.ly.out_chunk;
    const int t0 = tid >> 1, c0 = tid & 1;          // staging: lane pair = (token, chunk); pass u handles token t0 + u * THREADS / 2
    const float4 *org = img + WB + 1;               // token (y, x) = (0, 0) of the map inside the bordered image

    // ---- zero border (once: staging only ever writes the interior): rows 0, H+1, H+2 and columns 0, W+1 of rows 1..H ----
    {
        const int nb = 3 * WB + 2 * H + 1;          // + the token after the last row: (y1, x1) of a sample at (H, W)
        for (int j = tid; j < 2 * nb; j += THREADS) {
            const int c = j >= nb, k = j - c * nb;
            int tb;
            if (k < WB) tb = k;
            else if (k <= 3 * WB) tb = (H + 1) * WB + (k - WB);
            else if (k <= 3 * WB + H) tb = (k - 3 * WB) * WB;
            else tb = (k - 3 * WB - H) * WB + W + 1;
            img[c * PL + tb] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // item -> (frame, head, octet, query slice).  Items go round-robin to the workgroups and the grid is a multiple of 8
    // (or the item count), so the low 3 bits - the head - are fixed per workgroup and the 4 octets (and the query
    // slices) of one (frame, head) stay on one XCD / L2.
    float4 v[STAGE_PASSES];
    // staging loads of an item: lane pair = (token, chunk), 32 contiguous bytes per token; straight-line, per-lane
    // predicated: all loads in flight at once
    auto issue_stage = [&](int item) {
        const int head = item & 7, r = item >> 3;
        const int n = r / per, oct = (r - n * per) & 3;
        const float *__restrict__ vb = g.value + n * g.ly.value_frame + head * g.ly.value_head + oct * g.ly.value_oct;
        const float *src = vb + t0 * vs_token + c0 * vs_chunk;
        const long step = (THREADS / 2) * vs_token;
#pragma unroll
        for (int u = 0; u < STAGE_PASSES; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#if !(defined(DFX_LEVEL_ABLATE) && DFX_LEVEL_ABLATE == 4)       // (ablation 4, results wrong: the level is never loaded)
            if (t0 + u * (THREADS / 2) < S) v[u] = *reinterpret_cast<const float4 *>(src + u * step);
#endif
        }
    };
    // land the staged level in LDS: (y, x) of the pass's token advance by a constant step, no division
    auto land = [&]() {
        const int dy = (THREADS / 2) / W, dx = (THREADS / 2) - dy * W;
        int y = t0 / W, x = t0 - y * W;
        int idx = c0 * PL + (y + 1) * WB + x + 1;
#pragma unroll
        for (int u = 0; u < STAGE_PASSES; ++u) {
            if (t0 + u * (THREADS / 2) < S) img[idx] = v[u];
            x += dx;
            idx += dy * WB + dx;
            if (x >= W) { x -= W; idx += 2; }
        }
    };
    // ... and the same loads in slices of SL, one slice per gather iteration (PREFETCH): a wave's vector-memory counter retires
    // in order, so a wait for the next query's parameters also waits for every OLDER load - the whole image, if it were issued
    // in one go before the gather (measured: no gain at all).  A slice is issued right AFTER the next query's parameter loads:
    // that wait does not cover it, the wait one iteration later does, by when the slice has been in flight for a whole
    // iteration.
    constexpr int SL = 3, NSL = (STAGE_PASSES + SL - 1) / SL;
    auto stage_slice = [&](int s, const float *src) {            // s is wave-uniform: a scalar branch per slice
        const long step = (THREADS / 2) * vs_token;
#pragma unroll
        for (int ss = 0; ss < NSL; ++ss) {
            if (ss != s) continue;
#pragma unroll
            for (int u = ss * SL; u < (ss + 1) * SL && u < STAGE_PASSES; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t0 + u * (THREADS / 2) < S) v[u] = *reinterpret_cast<const float4 *>(src + u * step);
            }
        }
    };
    if (PREFETCH && (int)blockIdx.x < g.nitems) issue_stage(blockIdx.x);
    for (int item = blockIdx.x; item < g.nitems; item += gridDim.x) {
        const int head = item & 7, r = item >> 3;
        const int n = r / per, sub = r - n * per;
        const int oct = sub & 3, qs = sub >> 2;
        const int qbeg = qs * qper, qend = min(Lq, qbeg + qper);
        if (PREFETCH) {
            // the image of this item came in slice by slice under the previous item's gather: land it; the next item's
            // slices are issued inside the gather loop below
            land();
            __syncthreads();
        } else {
            issue_stage(item);
        }
        // ---- parameters and taps of the first query while the value loads fly ----
        const float *__restrict__ refn = g.ref + (long)n * Lq * REFDIM;
        const float *__restrict__ offn = g.off + (long)n * Lq * off_stride + head * g.ly.off_head;
        const float *__restrict__ lgn = g.logits + (long)n * Lq * logit_stride + head * g.ly.logit_head;
        float *__restrict__ outn = g.out + (long)n * Lq * g.ly.out_row + head * g.ly.out_head + oct * g.ly.out_oct;
        int q = qbeg + tid;
        bool have = q < qend;
        Raw raw;
        if (have) raw = load_raw<REFDIM>(refn + (long)q * REFDIM, offn + q * off_stride, lgn + q * logit_stride);
        Taps tp;
        if (have) tp = make_taps<REFDIM>(raw, lv);
        if (!PREFETCH) {
            land();
            __syncthreads();
        }

        // ---- gather: 16 corners x 2 chunks from LDS; the next query's parameters (and a slice of the next item's image)
        // load meanwhile.  The trip count is the workgroup's, not the thread's, so that the slice index stays uniform ----
        const int next_item = item + (int)gridDim.x;
        const bool more = PREFETCH && next_item < g.nitems;
        const float *src_next = nullptr;
        if (more) {
            const int nh = next_item & 7, nr = next_item >> 3, nn = nr / per, noct = (nr - nn * per) & 3;
            src_next = g.value + nn * g.ly.value_frame + nh * g.ly.value_head + noct * g.ly.value_oct + t0 * vs_token + c0 * vs_chunk;
        }
        int slice = 0;
        const int iters = (qend - qbeg + THREADS - 1) / THREADS;
        for (int it = 0; it < iters; ++it) {
            const int qn = q + THREADS;
            const bool hn = qn < qend;
            if (hn) raw = load_raw<REFDIM>(refn + (long)qn * REFDIM, offn + qn * off_stride, lgn + qn * logit_stride);
            if (more && slice < NSL) stage_slice(slice++, src_next);
            if (!have) continue;                                  // (only in a workgroup's last iteration)
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#if defined(DFX_LEVEL_ABLATE) && DFX_LEVEL_ABLATE == 3
            // timing ablation (results are wrong): no gather - the taps are computed and kept alive, nothing is read from LDS
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                a0.x += tp.wt[p].x + tp.wb[p].y;
                a1.x += tp.wt[p].y + tp.wb[p].x + (float)tp.tb[p];
            }
#else
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 *b0 = org + tp.tb[p];
                const float4 *b1 = b0 + PL;
                fma4(a0, tp.wt[p].x, b0[0]);
                fma4(a1, tp.wt[p].x, b1[0]);
                fma4(a0, tp.wt[p].y, b0[1]);
                fma4(a1, tp.wt[p].y, b1[1]);
                fma4(a0, tp.wb[p].x, b0[WB]);
                fma4(a1, tp.wb[p].x, b1[WB]);
                fma4(a0, tp.wb[p].y, b0[WB + 1]);
                fma4(a1, tp.wb[p].y, b1[WB + 1]);
            }
#endif
            float *dst = outn + q * out_row;
            *reinterpret_cast<float4 *>(dst) = a0;
            *reinterpret_cast<float4 *>(dst + out_chunk) = a1;
#if defined(DFX_LEVEL_ABLATE) && DFX_LEVEL_ABLATE == 1
            // timing ablation (tools/level_ablate.py; results are wrong): the tap arithmetic of every query after a thread's
            // first is skipped - its parameter loads stay (consumed by an empty asm) and the gather re-uses the first query's
            // taps - to price what sharing the taps between the 4 octet-workgroups could save
            if (hn) asm volatile("" :: "v"(raw.lg.x), "v"(raw.lg.w), "v"(raw.o01.x), "v"(raw.o01.w), "v"(raw.o23.x), "v"(raw.o23.w), "v"(raw.r.x));
#else
            if (hn) tp = make_taps<REFDIM>(raw, lv);
#endif
#if defined(DFX_LEVEL_ABLATE) && DFX_LEVEL_ABLATE == 2
            // timing ablation (results are wrong): neighbouring lanes read neighbouring tokens whatever their offsets - no LDS
            // bank conflict - to price the conflicts that per-query offsets cause
            if (hn) {
#pragma unroll
                for (int p = 0; p < 4; ++p) tp.tb[p] = (qn % (S - 8)) + p;
            }
#endif
            q = qn;
            have = hn;
        }
        if (more)
            for (; slice < NSL; ++slice) stage_slice(slice, src_next);      // (few queries per item: the rest of the image now)
        __syncthreads();                            // every gather of this item is done before the next level lands
    }
}

// plane stride in tokens: bordered tokens rounded up to 4 mod 8
inline long plane_tokens(int H, int W)
{
    const long nt = (long)(H + 3) * (W + 2);        // one token of border, one more row for y1 of dropped samples
    return nt + 1 + ((4 - (nt + 1) % 8) + 8) % 8;
}

}  // namespace

extern "C" int dfx_msda_fused_level_fits(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    return 2 * plane_tokens(H, W) * 16 <= LDS_CAP && (long)H * W * 2 <= (long)STAGE_SLOTS;
}

extern "C" int dfx_msda_fused_level_forward_f32(const float *value, const float *ref, int ref_dim, const float *off,
                                                const float *logits, const dfx_msda_level_layout *layout, int N,
                                                int H, int W, int Lq, float *out, void *stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Lq < 0) return dfx::fail(DFX_EINVAL, "msda level: bad dimension");
    if (N == 0 || Lq == 0) return DFX_OK;
    if (!value || !ref || !off || !logits || !out || !layout) return dfx::fail(DFX_EINVAL, "msda level: null pointer");
    if (ref_dim != 2 && ref_dim != 4) return dfx::fail(DFX_EINVAL, "msda level: ref_dim must be 2 or 4");
    const dfx_msda_level_layout &ly = *layout;
    const long all = ly.value_frame | ly.value_token | ly.value_head | ly.value_oct | ly.value_chunk | ly.off_row |
                     ly.off_head | ly.logit_row | ly.logit_head | ly.out_row | ly.out_head | ly.out_oct | ly.out_chunk;
    if ((all & 3) || all < 0 || ly.value_token < 4 || ly.value_chunk < 4 || ly.off_row < 8 || ly.logit_row < 4 ||
        ly.out_row < 4 || ly.out_chunk < 4)
        return dfx::fail(DFX_EINVAL, "msda level: strides must be positive multiples of 4 floats (16-byte vectors)");
    if (!dfx::aligned16(value) || !dfx::aligned16(out) || !dfx::aligned16(off) || !dfx::aligned16(logits) ||
        (ref_dim == 4 ? !dfx::aligned16(ref) : ((uintptr_t)ref & 7) != 0))
        return dfx::fail(DFX_EINVAL, "msda level: buffers must be 16-byte aligned");
    if (!dfx_msda_fused_level_fits(H, W))
        return dfx::fail(DFX_EINVAL, "msda level: a %d x %d level does not fit the 160 KB LDS image; "
                                     "use dfx_msda_fused_forward_f32", H, W);
    if ((long)N * Lq >= (1L << 28)) return dfx::fail(DFX_ERANGE, "msda level: too many queries");
    // DFX_LEVEL_VARIANT=1 selects the round-4 variant - 512 threads, the next item's level prefetched into registers under the
    // gather (see the kernel) - for A/B runs: measured 5-8 % SLOWER than the 1024-thread stage-then-gather schedule
    // (profiles/r04_level_variant.txt), which stays the default
    const bool pf = dfx::tuning().level_variant_set && dfx::tuning().level_variant == 1;
    const int threads = pf ? 512 : 1024;
    // enough workgroups for the 256 CUs: split the queries of a frame when there are few frames
    int qsplit = 1;
    while ((long)N * 32 * qsplit < 256 && qsplit < 8 && Lq / (qsplit * 2) >= 512) qsplit *= 2;
    const long blocks = (long)N * 32 * qsplit;
    if (blocks >= (1L << 31)) return dfx::fail(DFX_ERANGE, "msda level: grid too large");
    const int PL = (int)plane_tokens(H, W);
    const size_t lds = (size_t)2 * PL * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // LDS images above 64 KB need the per-function opt-in, once per DEVICE (the attribute is per device context)
    static std::mutex raise_mu;
    static bool raised_on[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> raise_lock(raise_mu);
    bool &raised = raised_on[dev & 63];
    if (!raised) {
        const void *fns[4] = {reinterpret_cast<const void *>(&msda_fused_level<2, 512, true>), reinterpret_cast<const void *>(&msda_fused_level<4, 512, true>),
                              reinterpret_cast<const void *>(&msda_fused_level<2, 1024, false>), reinterpret_cast<const void *>(&msda_fused_level<4, 1024, false>)};
        for (const void *fn : fns)
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP) != hipSuccess)
                return dfx::fail(DFX_ELAUNCH, "msda level: cannot raise the dynamic LDS limit");
        raised = true;
    }
    const int S = H * W;
    const LevelArgs g{value, ref, off, logits, out, ly, H, W, Lq, PL, qsplit, (int)blocks};
    // persistent: at most one workgroup per CU, a multiple of 8 so that item & 7 (the head) is fixed per workgroup
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        ncu = n / 8 * 8;
    }
    const long grid = (blocks < ncu || dfx::tuning().level_not_persistent) ? blocks : ncu;
    // algorithmic bytes of this launch: value + (offsets, logits) + out, fp32
    const long bytes = 4L * ((long)N * S * 256 + 3L * N * Lq * 8 * 4 + (long)N * Lq * 256);
    if (pf) {
        if (ref_dim == 2) dfx::launch_timed(bytes, Lq, S, msda_fused_level<2, 512, true>, dim3((unsigned)grid), dim3(threads), lds, st, g);
        else dfx::launch_timed(bytes, Lq, S, msda_fused_level<4, 512, true>, dim3((unsigned)grid), dim3(threads), lds, st, g);
    } else {
        if (ref_dim == 2) dfx::launch_timed(bytes, Lq, S, msda_fused_level<2, 1024, false>, dim3((unsigned)grid), dim3(threads), lds, st, g);
        else dfx::launch_timed(bytes, Lq, S, msda_fused_level<4, 1024, false>, dim3((unsigned)grid), dim3(threads), lds, st, g);
    }
    return dfx::check_launch("msda_fused_level");
}
