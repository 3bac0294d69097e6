// Sampling "taps" of the wave-per-query MSDA kernels (forward + fused front end), gfx950.
//
// A tap is one (query, head, level, point) sample reduced to what the gather needs:
//   4 byte offsets into the batch element's value slab (level start, corner row and the head's
//   128-byte column already folded in) and 4 bilinear weights already multiplied by the
//   attention weight (0 for corners outside the map and for samples the skip rule drops).
// Taps are computed ONCE per (query, head, point) by one lane - not redundantly by the 8 lanes
// that share a head - staged in LDS in [query][level][point][head] order, and read back by the
// gather phase as two conflict-free 16-byte broadcasts per point.
#pragma once
#include "dfx_common.h"

namespace dfx {

struct Tap {
    uint4 off;    // byte offsets of the corners (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    float4 w;     // matching weights
};

// Geometry of one sample; follows /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh
// :281-291 (pixel coordinates, skip rule) and :33-84 (corner validity, bilinear weights).
// `head_bytes` = m * 128, `level_row` = level_start_index[l] (rows of 1 KiB = 8 heads x 32 fp32).
__device__ __forceinline__ Tap make_tap(float lx, float ly, float a, int H, int W, int level_row,
                                        int head_bytes)
{
    const float h_im = ly * (float)H - 0.5f;
    const float w_im = lx * (float)W - 0.5f;
    const bool inr = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    // clamp in float first: keeps the float->int conversion defined for NaN / huge inputs
    const float hf = floorf(fminf(fmaxf(h_im, -1.f), (float)H));
    const float wf = floorf(fminf(fmaxf(w_im, -1.f), (float)W));
    const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool top = inr && h0 >= 0, bot = inr && h1 <= H - 1, lef = w0 >= 0, rig = w1 <= W - 1;
    Tap t;
    if (H <= 0 || W <= 0) {
        t.off = make_uint4(0u, 0u, 0u, 0u);
        t.w = make_float4(0.f, 0.f, 0.f, 0.f);
        return t;
    }
    t.w.x = (top && lef) ? hh * hw * a : 0.f;
    t.w.y = (top && rig) ? hh * lw * a : 0.f;
    t.w.z = (bot && lef) ? lh * hw * a : 0.f;
    t.w.w = (bot && rig) ? lh * lw * a : 0.f;
    // clamped, always-in-bounds addresses (an empty level, H or W == 0, never gets here: callers
    // give it zero weights and offset 0)
    const int y0 = max(min(h0, H - 1), 0), y1 = max(min(h1, H - 1), 0);
    const int x0 = max(min(w0, W - 1), 0), x1 = max(min(w1, W - 1), 0);
    const int r0 = level_row + y0 * W, r1 = level_row + y1 * W;
    t.off.x = (unsigned)(r0 + x0) * 1024u + (unsigned)head_bytes;
    t.off.y = (unsigned)(r0 + x1) * 1024u + (unsigned)head_bytes;
    t.off.z = (unsigned)(r1 + x0) * 1024u + (unsigned)head_bytes;
    t.off.w = (unsigned)(r1 + x1) * 1024u + (unsigned)head_bytes;
    return t;
}

__device__ __forceinline__ void fma4(float4 &acc, float w, const float4 &v)
{
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z);
    acc.w = fmaf(w, v.w, acc.w);
}

// Orders a wave's own LDS writes before its own later LDS reads (other lanes' data).  The LDS
// executes one wave's instructions in order, so this only has to stop the compiler from moving
// accesses across it.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gather phase for ONE query: lane = (head m = lane&7 ... see callers), reads its head's taps of
// every (level, point) from LDS and accumulates 4 channels.
//   vb        : wave-uniform pointer to the batch element's value slab
//   lane_b    : cg * 16 (byte offset of this lane's channel quad inside the head's 128 bytes)
//   toff/tw   : LDS tap arrays of this query, [LT][4 points][8 heads]
template <int LT>
__device__ __forceinline__ float4 gather_query(const char *__restrict__ vb, unsigned lane_b, int m,
                                               const uint4 *toff, const float4 *tw)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        uint4 o[4];
        float4 w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            o[p] = toff[(l * 4 + p) * 8 + m];
            w[p] = tw[(l * 4 + p) * 8 + m];
        }
        float4 v[16];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p * 4 + 0] = *reinterpret_cast<const float4 *>(vb + (o[p].x + lane_b));
            v[p * 4 + 1] = *reinterpret_cast<const float4 *>(vb + (o[p].y + lane_b));
            v[p * 4 + 2] = *reinterpret_cast<const float4 *>(vb + (o[p].z + lane_b));
            v[p * 4 + 3] = *reinterpret_cast<const float4 *>(vb + (o[p].w + lane_b));
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            fma4(acc, w[p].x, v[p * 4 + 0]);
            fma4(acc, w[p].y, v[p * 4 + 1]);
            fma4(acc, w[p].z, v[p * 4 + 2]);
            fma4(acc, w[p].w, v[p * 4 + 3]);
        }
    }
    return acc;
}

}  // namespace dfx
