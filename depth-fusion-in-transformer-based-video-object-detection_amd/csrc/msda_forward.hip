// Multi-scale deformable attention, forward, for gfx950 (MI355X / CDNA4).
//
// Replaces /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (kernel)
// and ms_deform_attn_cuda.cu:20-80 (host) behind the C ABI in include/dfx_msda.h.
//
// The reference gives one THREAD one output channel (1024-thread blocks, int64 shape loads and
// scalar 4-byte gathers per thread, every thread redoing the sample geometry).  Here the unit of
// work is a WAVE and the work is split in two phases (production geometry M=8 heads x D=32
// channels, P=4 points, fp32):
//
//   phase A  "taps": lane = one (query, level, point, head) sample of the wave's 2 queries.  It
//            loads its (x, y, weight) triple - the 64 lanes read 768 contiguous bytes - and does
//            the geometry ONCE: skip rule, floor, corner validity, 4 byte offsets, 4 weights
//            premultiplied by the attention weight.  Taps go to LDS in [query][level][point][head]
//            order (32 bytes each; two 16-byte stores).
//   phase B  "gather": lane = (head m = lane>>3, channel quad cg = lane&7).  Per point it reads
//            its head's tap back as two 16-byte LDS broadcasts (8 distinct addresses, 128
//            contiguous bytes: conflict-free), then issues the 4 corner fetches as 16-byte loads
//            with a wave-uniform base + 32-bit offset; the 8 lanes of a head read one contiguous
//            128-byte value row, so a wave-instruction touches 8 rows.  16 gathers per level are
//            in flight before the first FMA; the reduction over L*P samples stays in registers
//            and the query's 1 KiB output row leaves as one float4 per lane.
//
// A 256-thread workgroup holds 4 waves x 2 queries = 8 consecutive queries (x `iters`);
// workgroups are remapped so that each XCD walks a contiguous raster range of queries and its
// private L2 holds only that band of the value map (dfx_common.h).  The earlier one-phase form
// (every lane redoing the geometry) measured VALU-bound at ~1000 instructions per query; it is
// kept below only for P != 4.
//
// Roofline: HBM-bound gather.  Algorithmic bytes per call
//   4 * (N*S*M*D  +  3*N*Lq*M*L*P  +  N*Lq*M*D)       (value + loc/aw + out, fp32)
// = 10.21 MB per frame for the encoder geometry (S = Lq = 4200, L = 1).
#include "dfx_common.h"
#include "msda_tap.h"

namespace {

using dfx::xcd_remap;

struct Corner4 {
    int o00, o01, o10, o11;      // element offsets (in floats) of the 4 corners inside one level
    float w00, w01, w10, w11;    // bilinear weights, already multiplied by the attention weight
};

// Geometry of one sample.  Follows ms_deform_im2col_cuda.cuh:281-291 (pixel coords, skip rule)
// and :33-84 (corner validity, weights).  Invalid corners / skipped samples get weight 0 and a
// clamped (always in-bounds) address.
__device__ __forceinline__ Corner4 corners(float lx, float ly, float a, int H, int W, int row_stride)
{
    const float h_im = ly * (float)H - 0.5f;
    const float w_im = lx * (float)W - 0.5f;
    const bool inr = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    // clamp in float first: keeps the float->int conversion defined for NaN / huge inputs
    const float hf = floorf(fminf(fmaxf(h_im, -1.f), (float)H));
    const float wf = floorf(fminf(fmaxf(w_im, -1.f), (float)W));
    const int h0 = (int)hf, w0 = (int)wf;
    const int h1 = h0 + 1, w1 = w0 + 1;
    const float lh = h_im - hf, lw = w_im - wf;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool top = inr && h0 >= 0, bot = inr && h1 <= H - 1, lef = w0 >= 0, rig = w1 <= W - 1;
    Corner4 c;
    c.w00 = (top && lef) ? hh * hw * a : 0.f;
    c.w01 = (top && rig) ? hh * lw * a : 0.f;
    c.w10 = (bot && lef) ? lh * hw * a : 0.f;
    c.w11 = (bot && rig) ? lh * lw * a : 0.f;
    const int y0 = min(max(h0, 0), H - 1), y1 = min(max(h1, 0), H - 1);
    const int x0 = min(max(w0, 0), W - 1), x1 = min(max(w1, 0), W - 1);
    c.o00 = (y0 * W + x0) * row_stride;
    c.o01 = (y0 * W + x1) * row_stride;
    c.o10 = (y1 * W + x0) * row_stride;
    c.o11 = (y1 * W + x1) * row_stride;
    return c;
}

using dfx::fma4;
using dfx::Tap;

// ---------------------------------------------------------------------------------------------
// Fast path: M = 8, D = 32, fp32, 16-byte aligned buffers.  PT = points per level at compile
// time (4 in every shipped config); PT = 0 keeps P a run-time value (scalar loc/aw loads).
// ---------------------------------------------------------------------------------------------
template <int PT, bool REMAP>
__global__ __launch_bounds__(256) void msda_fwd_m8d32(const float *__restrict__ value,
                                                      const int64_t *__restrict__ shapes,
                                                      const int64_t *__restrict__ lsi,
                                                      const float *__restrict__ loc,
                                                      const float *__restrict__ aw,
                                                      int NQ, int Lq, int S, int L, int Prt,
                                                      float *__restrict__ out)
{
    const int blk = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int qi = blk * 4 + (threadIdx.x >> 6);   // flat query index over N*Lq
    if (qi >= NQ) return;
    const int P = PT ? PT : Prt;
    const int m = lane >> 3, cg = lane & 7;
    const int b = qi / Lq;
    const long samp = (long)qi * 8 + m;            // flat (b,q,m) index
    const float *vb = value + (long)b * S * 256 + m * 32 + cg * 4;
    const float *lp = loc + samp * (long)(L * P * 2);
    const float *ap = aw + samp * (long)(L * P);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float *vl = vb + (long)((int)lsi[l]) * 256;
        if (H <= 0 || W <= 0) continue;   // empty level: nothing can be in range
        if (PT == 4) {
            const float4 la = *reinterpret_cast<const float4 *>(lp + l * 8);
            const float4 lb = *reinterpret_cast<const float4 *>(lp + l * 8 + 4);
            const float4 a4 = *reinterpret_cast<const float4 *>(ap + l * 4);
            const Corner4 c0 = corners(la.x, la.y, a4.x, H, W, 256);
            const Corner4 c1 = corners(la.z, la.w, a4.y, H, W, 256);
            const Corner4 c2 = corners(lb.x, lb.y, a4.z, H, W, 256);
            const Corner4 c3 = corners(lb.z, lb.w, a4.w, H, W, 256);
#define DFX_LD(c, o) (*reinterpret_cast<const float4 *>(vl + c.o))
            const float4 v00 = DFX_LD(c0, o00), v01 = DFX_LD(c0, o01), v02 = DFX_LD(c0, o10), v03 = DFX_LD(c0, o11);
            const float4 v10 = DFX_LD(c1, o00), v11 = DFX_LD(c1, o01), v12 = DFX_LD(c1, o10), v13 = DFX_LD(c1, o11);
            const float4 v20 = DFX_LD(c2, o00), v21 = DFX_LD(c2, o01), v22 = DFX_LD(c2, o10), v23 = DFX_LD(c2, o11);
            const float4 v30 = DFX_LD(c3, o00), v31 = DFX_LD(c3, o01), v32 = DFX_LD(c3, o10), v33 = DFX_LD(c3, o11);
#undef DFX_LD
            fma4(acc, c0.w00, v00); fma4(acc, c0.w01, v01); fma4(acc, c0.w10, v02); fma4(acc, c0.w11, v03);
            fma4(acc, c1.w00, v10); fma4(acc, c1.w01, v11); fma4(acc, c1.w10, v12); fma4(acc, c1.w11, v13);
            fma4(acc, c2.w00, v20); fma4(acc, c2.w01, v21); fma4(acc, c2.w10, v22); fma4(acc, c2.w11, v23);
            fma4(acc, c3.w00, v30); fma4(acc, c3.w01, v31); fma4(acc, c3.w10, v32); fma4(acc, c3.w11, v33);
        } else {
            for (int p = 0; p < P; ++p) {
                const float lx = lp[(l * P + p) * 2], ly = lp[(l * P + p) * 2 + 1];
                const Corner4 c = corners(lx, ly, ap[l * P + p], H, W, 256);
                const float4 v0 = *reinterpret_cast<const float4 *>(vl + c.o00);
                const float4 v1 = *reinterpret_cast<const float4 *>(vl + c.o01);
                const float4 v2 = *reinterpret_cast<const float4 *>(vl + c.o10);
                const float4 v3 = *reinterpret_cast<const float4 *>(vl + c.o11);
                fma4(acc, c.w00, v0); fma4(acc, c.w01, v1); fma4(acc, c.w10, v2); fma4(acc, c.w11, v3);
            }
        }
    }
    *reinterpret_cast<float4 *>(out + (long)qi * 256 + m * 32 + cg * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// Fast path: M = 8, D = 32, P = 4, fp32, LT levels (1..4) known at compile time.
// ---------------------------------------------------------------------------------------------
template <int LT>
__global__ __launch_bounds__(256) void msda_fwd_taps(const float *__restrict__ value,
                                                     const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lsi,
                                                     const float *__restrict__ loc,
                                                     const float *__restrict__ aw, int NQ, int Lq,
                                                     int S, int iters, float *__restrict__ out)
{
    constexpr int QW = 2;                 // queries per wave per iteration
    constexpr int TAPS = QW * LT * 32;    // taps per wave per iteration (multiple of 64)
    __shared__ uint4 s_off[4][TAPS];
    __shared__ float4 s_w[4][TAPS];
    // wave index as a scalar: everything derived from it (query index, batch element, the value
    // slab pointer) then lives in SGPRs and the gathers use scalar-base + 32-bit-offset addressing
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint4 *toff = s_off[wave];
    float4 *tw = s_w[wave];
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int m = lane >> 3;
    const unsigned lane_b = (unsigned)(lane & 7) * 16u;

    int Hs[LT], Ws[LT], Rs[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        Hs[l] = (int)shapes[2 * l];
        Ws[l] = (int)shapes[2 * l + 1];
        Rs[l] = (int)lsi[l];
    }

    for (int it = 0; it < iters; ++it) {
        const int q0 = ((blk * iters + it) * 4 + wave) * QW;   // first query of this wave (uniform)
        if (q0 >= NQ) break;
        // ---- phase A: one tap per lane ----
#pragma unroll
        for (int c = 0; c < TAPS / 64; ++c) {
            const int s = c * 64 + lane;                 // slot = ((qq*LT + l)*4 + p)*8 + head
            const int hm = s & 7, p = (s >> 3) & 3, ql = s >> 5;
            const int l = (LT == 1) ? 0 : ql % LT, qq = (LT == 1) ? ql : ql / LT;
            const int qi = q0 + qq;
            Tap t;
            if (qi < NQ) {
                const long e = (((long)qi * 8 + hm) * LT + l) * 4 + p;
                const float2 xy = *reinterpret_cast<const float2 *>(loc + e * 2);
                int H = Hs[0], W = Ws[0], R = Rs[0];
#pragma unroll
                for (int k = 1; k < LT; ++k)
                    if (l == k) { H = Hs[k]; W = Ws[k]; R = Rs[k]; }
                t = dfx::make_tap(xy.x, xy.y, aw[e], H, W, R, hm * 128);
            } else {
                t.off = make_uint4(0u, 0u, 0u, 0u);
                t.w = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            toff[s] = t.off;
            tw[s] = t.w;
        }
        dfx::wave_lds_fence();
        // ---- phase B: gather, one query at a time ----
#pragma unroll
        for (int qq = 0; qq < QW; ++qq) {
            const int qi = q0 + qq;
            if (qi < NQ) {
                const int b = qi / Lq;
                const char *vb = reinterpret_cast<const char *>(value) + (size_t)b * S * 1024;
                const float4 acc = dfx::gather_query<LT>(vb, lane_b, m, toff + qq * LT * 32, tw + qq * LT * 32);
                *reinterpret_cast<float4 *>(out + (long)qi * 256 + lane * 4) = acc;
            }
        }
        dfx::wave_lds_fence();   // the next iteration overwrites the taps
    }
}

// ---------------------------------------------------------------------------------------------
// Generic path: any M, D, L, P; fp32 and fp64.  One thread per output element, channel fastest
// (adjacent lanes read adjacent channels of the same value row), grid-stride.  Used by the
// reference's tiny test fixture (M=D=2), odd head sizes and every fp64 call.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic(const T *__restrict__ value,
                                                        const int64_t *__restrict__ shapes,
                                                        const int64_t *__restrict__ lsi,
                                                        const T *__restrict__ loc,
                                                        const T *__restrict__ aw, long total, int S,
                                                        int M, int D, int L, int Lq, int P,
                                                        T *__restrict__ out)
{
    const int row = M * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        long t = idx;
        const int c = (int)(t % D);
        t /= D;
        const long samp = t;
        const int m = (int)(t % M);
        t /= M;
        const int b = (int)(t / Lq);
        const T *vb = value + (long)b * S * row + m * D + c;
        long wp = samp * L * P, lp = wp * 2;
        T col = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const T *vl = vb + (long)((int)lsi[l]) * row;
            for (int p = 0; p < P; ++p, ++wp, lp += 2) {
                const T h_im = loc[lp + 1] * (T)H - (T)0.5;
                const T w_im = loc[lp] * (T)W - (T)0.5;
                if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
                    const T hf = floor(h_im), wf = floor(w_im);
                    const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
                    const T lh = h_im - hf, lw = w_im - wf, hh = (T)1 - lh, hw = (T)1 - lw;
                    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                    if (h0 >= 0 && w0 >= 0) v1 = vl[(long)(h0 * W + w0) * row];
                    if (h0 >= 0 && w1 <= W - 1) v2 = vl[(long)(h0 * W + w1) * row];
                    if (h1 <= H - 1 && w0 >= 0) v3 = vl[(long)(h1 * W + w0) * row];
                    if (h1 <= H - 1 && w1 <= W - 1) v4 = vl[(long)(h1 * W + w1) * row];
                    col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * aw[wp];
                }
            }
        }
        out[idx] = col;
    }
}

}  // namespace

extern "C" int dfx_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                                    const float *loc, const float *aw, int N, int S, int M, int D,
                                    int L, int Lq, int P, float *out, void *stream)
{
    const int rc = dfx::check_dims(value, shapes, lsi, loc, aw, out, N, S, M, D, L, Lq, P);
    if (rc < 0) return rc;
    if (rc == 1) return DFX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long nq = (long)N * Lq;
    if (S == 0 || L == 0 || P == 0) {   // nothing to sample: the reference returns zeros
        if (hipMemsetAsync(out, 0, sizeof(float) * nq * M * D, st) != hipSuccess)
            return dfx::fail(DFX_ELAUNCH, "msda forward: memset failed");
        return DFX_OK;
    }
    const bool fast = M == 8 && D == 32 && nq < (1L << 28) && (long)S * 1024 < (1L << 32) &&
                      dfx::aligned16(value) && dfx::aligned16(out);
    if (fast && P == 4 && L <= 4 && (reinterpret_cast<uintptr_t>(loc) & 7u) == 0) {
        // 8 queries per workgroup and iteration; keep >= ~2048 workgroups in the grid when we can
        int iters = 1;
        while (iters < 8 && nq / (8L * iters * 2) >= 2048) iters *= 2;
        const int grid = (int)((nq + 8L * iters - 1) / (8L * iters));
#define DFX_LAUNCH(LT)                                                                              \
        hipLaunchKernelGGL((msda_fwd_taps<LT>), dim3(grid), dim3(256), 0, st, value, shapes, lsi, loc, \
                           aw, (int)nq, Lq, S, iters, out)
        switch (L) {
            case 1: DFX_LAUNCH(1); break;
            case 2: DFX_LAUNCH(2); break;
            case 3: DFX_LAUNCH(3); break;
            default: DFX_LAUNCH(4); break;
        }
#undef DFX_LAUNCH
        return dfx::check_launch("msda_fwd_taps");
    }
    if (fast) {
        const int grid = (int)((nq + 3) / 4);
        hipLaunchKernelGGL((msda_fwd_m8d32<0, true>), dim3(grid), dim3(256), 0, st, value, shapes, lsi,
                           loc, aw, (int)nq, Lq, S, L, P, out);
        return dfx::check_launch("msda_fwd_m8d32");
    }
    const long total = nq * M * D;
    hipLaunchKernelGGL((msda_fwd_generic<float>), dim3(dfx::grid_for(total)), dim3(256), 0, st, value, shapes, lsi, loc,
                       aw, total, S, M, D, L, Lq, P, out);
    return dfx::check_launch("msda_fwd_generic<float>");
}

extern "C" int dfx_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                                    const double *loc, const double *aw, int N, int S, int M, int D,
                                    int L, int Lq, int P, double *out, void *stream)
{
    const int rc = dfx::check_dims(value, shapes, lsi, loc, aw, out, N, S, M, D, L, Lq, P);
    if (rc < 0) return rc;
    if (rc == 1) return DFX_OK;
    const long total = (long)N * Lq * M * D;
    hipLaunchKernelGGL((msda_fwd_generic<double>), dim3(dfx::grid_for(total)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), value, shapes, lsi, loc, aw, total, S, M, D, L,
                       Lq, P, out);
    return dfx::check_launch("msda_fwd_generic<double>");
}

extern "C" int dfx_abi_version(void) { return 3; }

extern "C" int dfx_profile_enable(int on)
{
    dfx::profile_state().enabled = on != 0;
    return DFX_OK;
}

extern "C" int dfx_profile_drain(float *ms, long *bytes, int *tag_a, int *tag_b, int cap)
{
    dfx::ProfileState &p = dfx::profile_state();
    std::lock_guard<std::mutex> lock(p.mu);
    int n = 0;
    for (auto &r : p.records) {
        float t = -1.f;
        if (hipEventSynchronize(r.stop) == hipSuccess) (void)hipEventElapsedTime(&t, r.start, r.stop);
        if (n < cap) {
            if (ms) ms[n] = t;
            if (bytes) bytes[n] = r.bytes;
            if (tag_a) tag_a[n] = r.tag_a;
            if (tag_b) tag_b[n] = r.tag_b;
            ++n;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    p.records.clear();
    return n;
}

extern "C" const char *dfx_last_error(void) { return dfx::err_slot(); }

extern "C" int dfx_tuning_reload(void)
{
    dfx::tuning_slot().read();
    return DFX_OK;
}
