// Fused MSDA for the encoder / depth-fusion geometry, LDS-tiled (gfx950).
//
// Geometry: one level (L = 1), queries in raster order over the same H x W grid as the value map
// (Lq == H*W): encoder self-attention, Late Fusion, Encoder Cross Fusion, the backbone fusion block.
// There every query samples around its own pixel, and neighbouring queries sample neighbouring
// value rows.  The wave-per-query kernel (msda_fused.hip) fetches every corner from L2: PMC
// counters show ~550 MB of L2 requests per 8-frame launch for 47.6 MB of compulsory reads (L1 hit
// rate ~0: 32 KiB of L1 against ~3.5k lines in flight per CU), i.e. it runs at the L2 gather
// rate (~21 TB/s), not at the HBM rate.  This kernel moves the reuse into LDS:
//
//   workgroup = one tile of TH x TW queries of one frame, ONE head (256 threads, thread = query)
//   stage   the head's 128-byte value rows of the tile plus a halo (4 px up/left, 5 px down/right)
//           once into LDS with coalesced 16-byte loads: (TH+9) x (TW+9) x 128 B (73 KB for 10 x 21)
//   sample  thread = (query, head): softmax of its 4 logits, 4 locations, 16 bilinear corners;
//           every corner reads its 128-byte row from LDS as 8 x ds_read_b128 (chunk index swizzled
//           by the row index so the 16 lanes of a read group hit 16 distinct 16-byte bank slots)
//           and accumulates all 32 channels of the head in registers; a corner that falls outside
//           the staged window (large learned offsets) is fetched from global memory instead, so
//           the result never depends on the tile shape
//   store   each thread writes its 128 contiguous output bytes.
//
// L2/HBM traffic per launch drops from "16 rows per (query, head)" to "(window / tile) rows per
// (query, head)" = 2.7x the value map for the 10 x 21 tile.  Heads are the fastest-varying part
// of the block index, so under the round-robin XCD placement each XCD's L2 sees one head's
// 128-byte column of the value rows.
#include "dfx_common.h"
#include <stdlib.h>

namespace {

constexpr int HALO_LO = 4;   // rows/columns staged before the tile
constexpr int HALO_HI = 5;   // ... and after it (bilinear needs x0+1)

__device__ __forceinline__ void fma4(float4 &a, float w, const float4 &v)
{
    a.x = fmaf(w, v.x, a.x);
    a.y = fmaf(w, v.y, a.y);
    a.z = fmaf(w, v.z, a.z);
    a.w = fmaf(w, v.w, a.w);
}

template <int REFDIM>
__global__ __launch_bounds__(256, 2) void msda_fused_tile(const float *__restrict__ value,
                                                          const float *__restrict__ ref,
                                                          const float *__restrict__ off, long off_stride,
                                                          const float *__restrict__ logits, long logit_stride,
                                                          int H, int W, int TH, int TW, int tiles_x, int tiles_per_frame,
                                                          float *__restrict__ out)
{
    extern __shared__ float4 win[];                 // [WH*WW positions][8 chunks], chunk-swizzled
    const int tid = threadIdx.x;
    const int h = blockIdx.x & 7;                   // head: fastest, so an XCD keeps one head
    const int t = blockIdx.x >> 3;
    const int b = t / tiles_per_frame;
    const int tile = t - b * tiles_per_frame;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const int wy0 = ty0 - HALO_LO, wx0 = tx0 - HALO_LO;
    const int WH = TH + HALO_LO + HALO_HI, WW = TW + HALO_LO + HALO_HI;
    const long S = (long)H * W;
    const float *vb = value + (long)b * S * 256 + h * 32;      // this frame, this head

    // ---- stage the window: 8 threads per position, 32 positions per pass; ALL passes (<= 20, the
    //      window is capped at 640 positions) are issued before the first LDS store, so a workgroup
    //      pays one memory round trip for its window, not one per pass ----
    {
        const int c = tid & 7;
        const int npos = WH * WW;
        constexpr int PASSES = 20;
        float4 v[PASSES];
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int ip = (tid >> 3) + u * 32;
            const int py = wy0 + ip / WW, px = wx0 + ip % WW;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ip < npos && py >= 0 && py < H && px >= 0 && px < W)
                v[u] = *reinterpret_cast<const float4 *>(vb + ((long)py * W + px) * 256 + c * 4);
        }
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int ip = (tid >> 3) + u * 32;
            if (ip < npos) win[ip * 8 + ((c + (ip >> 1)) & 7)] = v[u];
        }
    }
    __syncthreads();

    // ---- one (query, head) per thread ----
    const int qy = ty0 + tid / TW, qx = tx0 + tid % TW;
    if (tid >= TH * TW || qy >= H || qx >= W) return;
    const long qi = (long)b * S + (long)qy * W + qx;

    const float4 lg = *reinterpret_cast<const float4 *>(logits + qi * logit_stride + h * 4);
    const float mx = fmaxf(fmaxf(lg.x, lg.y), fmaxf(lg.z, lg.w));
    float e[4] = {expf(lg.x - mx), expf(lg.y - mx), expf(lg.z - mx), expf(lg.w - mx)};
    float sum = 0.f;
    sum += e[0]; sum += e[1]; sum += e[2]; sum += e[3];
    const float4 oa = *reinterpret_cast<const float4 *>(off + qi * off_stride + h * 8);
    const float4 ob = *reinterpret_cast<const float4 *>(off + qi * off_stride + h * 8 + 4);
    const float ox[4] = {oa.x, oa.z, ob.x, ob.z}, oy[4] = {oa.y, oa.w, ob.y, ob.w};
    float rx, ry, rw = 0.f, rh = 0.f;
    if (REFDIM == 2) {
        rx = ref[qi * 2];
        ry = ref[qi * 2 + 1];
    } else {
        const float4 rr = *reinterpret_cast<const float4 *>(ref + qi * 4);
        rx = rr.x; ry = rr.y; rw = rr.z; rh = rr.w;
    }

    float4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    // geometry of the 4 points first: weights, clamped corner coordinates, and whether every
    // corner of this lane lies inside the staged window
    float cw[4][4];
    int cyx[4][4];          // window-relative (iy << 16 | ix) when inside, absolute (cy << 16 | cx) otherwise
    bool inside = true;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float lx, ly;
        if (REFDIM == 2) {                       // ref + off / (W, H)      (ms_deform_attn.py:102-107)
            lx = rx + ox[p] / (float)W;
            ly = ry + oy[p] / (float)H;
        } else {                                 // ref_xy + off / P * ref_wh * 0.5   (:108-110)
            lx = rx + ox[p] / 4.f * rw * 0.5f;
            ly = ry + oy[p] / 4.f * rh * 0.5f;
        }
        const float a = e[p] / sum;
        // ms_deform_im2col_cuda.cuh:281-291 (skip rule) and :33-84 (corners)
        const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
        const bool inr = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
        const float hf = floorf(fminf(fmaxf(h_im, -1.f), (float)H));
        const float wf = floorf(fminf(fmaxf(w_im, -1.f), (float)W));
        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool top = inr && h0 >= 0, bot = inr && h1 <= H - 1, lef = w0 >= 0, rig = w1 <= W - 1;
        cw[p][0] = (top && lef) ? hh * hw * a : 0.f;
        cw[p][1] = (top && rig) ? hh * lw * a : 0.f;
        cw[p][2] = (bot && lef) ? lh * hw * a : 0.f;
        cw[p][3] = (bot && rig) ? lh * lw * a : 0.f;
        const int y0 = max(min(h0, H - 1), 0), y1 = max(min(h1, H - 1), 0);
        const int x0 = max(min(w0, W - 1), 0), x1 = max(min(w1, W - 1), 0);
        const bool in_p = (y0 >= wy0) && (y1 < wy0 + WH) && (x0 >= wx0) && (x1 < wx0 + WW);
        inside = inside && in_p;
        cyx[p][0] = (y0 << 16) | x0; cyx[p][1] = (y0 << 16) | x1;
        cyx[p][2] = (y1 << 16) | x0; cyx[p][3] = (y1 << 16) | x1;
    }

    if (__all(inside)) {
        // fast path, whole wave: every corner is staged - unconditional LDS reads, no branches
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ip = ((cyx[p][c] >> 16) - wy0) * WW + ((cyx[p][c] & 0xffff) - wx0);
                const float4 *row = win + ip * 8;
                const int sw = ip >> 1;
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = row[(k + sw) & 7];
#pragma unroll
                for (int k = 0; k < 8; ++k) fma4(acc[k], cw[p][c], v[k]);
            }
        }
    } else {
        // some lane samples outside its tile's window (large learned offsets): per-corner choice
        for (int p = 0; p < 4; ++p) {
            for (int c = 0; c < 4; ++c) {
                const int gy = cyx[p][c] >> 16, gx = cyx[p][c] & 0xffff;
                const int iy = gy - wy0, ix = gx - wx0;
                const float wgt = cw[p][c];
                if (iy >= 0 && iy < WH && ix >= 0 && ix < WW) {
                    const int ip = iy * WW + ix;
                    const float4 *row = win + ip * 8;
                    const int sw = ip >> 1;
#pragma unroll
                    for (int k = 0; k < 8; ++k) fma4(acc[k], wgt, row[(k + sw) & 7]);
                } else if (wgt != 0.f) {
                    const float4 *row = reinterpret_cast<const float4 *>(vb + ((long)gy * W + gx) * 256);
#pragma unroll
                    for (int k = 0; k < 8; ++k) fma4(acc[k], wgt, row[k]);
                }
            }
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(out + qi * 256 + h * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = acc[k];
}

// tile shape: maximise (queries actually covered / threads) x (tile area / window area)
void pick_tile(int H, int W, int &TH, int &TW, long cap_bytes)
{
    double best = -1.0;
    TH = 8; TW = 32;
    for (int th = 4; th <= 32; ++th) {
        int tw = 256 / th;
        if (tw > W) tw = W;
        if (tw < 4) continue;
        const long win_bytes = (long)(th + HALO_LO + HALO_HI) * (tw + HALO_LO + HALO_HI) * 128;
        if (win_bytes > cap_bytes) continue;
        const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
        const double util = (double)H * W / ((double)ty * tx * 256.0);
        const double reuse = (double)th * tw / ((double)(th + 9) * (tw + 9));
        const double score = util * reuse;
        if (score > best) { best = score; TH = th; TW = tw; }
    }
    // also try widths that divide W exactly (fewer ragged tiles)
    for (int tw = 8; tw <= W && tw <= 64; ++tw) {
        if (W % tw) continue;
        for (int th = 4; th <= 32 && th * tw <= 256; ++th) {
            const long win_bytes = (long)(th + 9) * (tw + 9) * 128;
            if (win_bytes > cap_bytes) continue;
            const int ty = (H + th - 1) / th, tx = W / tw;
            const double util = (double)H * W / ((double)ty * tx * 256.0);
            const double reuse = (double)th * tw / ((double)(th + 9) * (tw + 9));
            if (util * reuse > best) { best = util * reuse; TH = th; TW = tw; }
        }
    }
}

}  // namespace

extern "C" int dfx_msda_fused_tile_forward_f32(const float *value, const float *ref, int ref_dim, const float *off,
                                               long off_stride, const float *logits, long logit_stride, int N,
                                               int H, int W, float *out, void *stream)
{
    if (N < 0 || H <= 0 || W <= 0) return dfx::fail(DFX_EINVAL, "msda tile: bad dimension");
    if (N == 0) return DFX_OK;
    if (!value || !ref || !off || !logits || !out) return dfx::fail(DFX_EINVAL, "msda tile: null pointer");
    if (ref_dim != 2 && ref_dim != 4) return dfx::fail(DFX_EINVAL, "msda tile: ref_dim must be 2 or 4");
    if (off_stride < 64 || logit_stride < 32 || (off_stride & 3) || (logit_stride & 3))
        return dfx::fail(DFX_EINVAL, "msda tile: bad row strides");
    if (!dfx::aligned16(value) || !dfx::aligned16(out) || !dfx::aligned16(off) || !dfx::aligned16(logits) ||
        (ref_dim == 4 && !dfx::aligned16(ref)))
        return dfx::fail(DFX_EINVAL, "msda tile: buffers must be 16-byte aligned");
    if ((long)N * H * W >= (1L << 28)) return dfx::fail(DFX_ERANGE, "msda tile: too many queries");
    int TH, TW;
    const char *cap_env = getenv("DFX_TILE_LDS_KB");
    const long cap = (cap_env ? atol(cap_env) : 80) * 1024;
    pick_tile(H, W, TH, TW, cap);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long blocks = (long)N * tiles_x * tiles_y * 8;
    if (blocks >= (1L << 31)) return dfx::fail(DFX_ERANGE, "msda tile: grid too large");
    const size_t lds = (size_t)(TH + HALO_LO + HALO_HI) * (TW + HALO_LO + HALO_HI) * 128;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // windows above the default dynamic-LDS limit need the per-function opt-in (once per process)
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_tile<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fused_tile<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return dfx::fail(DFX_ELAUNCH, "msda tile: cannot raise the dynamic LDS limit");
        raised = true;
    }
    if (ref_dim == 2)
        hipLaunchKernelGGL((msda_fused_tile<2>), dim3((unsigned)blocks), dim3(256), lds, st, value, ref, off, off_stride,
                           logits, logit_stride, H, W, TH, TW, tiles_x, tiles_x * tiles_y, out);
    else
        hipLaunchKernelGGL((msda_fused_tile<4>), dim3((unsigned)blocks), dim3(256), lds, st, value, ref, off, off_stride,
                           logits, logit_stride, H, W, TH, TW, tiles_x, tiles_x * tiles_y, out);
    return dfx::check_launch("msda_fused_tile");
}
