// Multi-scale deformable attention, backward, for gfx950 (MI355X / CDNA4).
//
// Replaces ms_deform_attn_cuda_backward (/root/reference/models/ops/src/cuda/ms_deform_attn_cuda.cu:83-153)
// and the six col2im kernel variants of ms_deform_im2col_cuda.cuh:301-920.  The reference picks a
// kernel by channel count because it reduces grad_sampling_loc / grad_attn_weight over the D
// channels of one (query, head) through LDS (serial or tree reduce, block = D threads).
//
// Here the production geometry (M=8, D=32, fp32) keeps the forward's mapping - one wave per query,
// lane = (head, channel quad) - so the 32 channels of a head live in 8 adjacent lanes x 4 registers
// and the reduction is 3 in-register adds + 3 DPP/xor shuffles; no LDS, no barrier.  grad_value is
// scattered with hardware float atomics (global_atomic_add_f32: 4 per corner per lane, 128
// contiguous bytes per head row).  Every other geometry (and fp64) takes a thread-per-element
// path that accumulates all three gradients with atomics, like the reference's `_gm` variant
// (ms_deform_im2col_cuda.cuh:845-920).  All three outputs are accumulated into: the caller
// zero-fills them first (the reference allocates them with zeros_like).
#include "dfx_common.h"

namespace {

using dfx::xcd_remap;

template <typename T>
__device__ __forceinline__ void atomic_add(T *p, T v)
{
    unsafeAtomicAdd(p, v);   // hardware global_atomic_add_{f32,f64}, no CAS loop
}

__device__ __forceinline__ float head_sum(float v)
{
    // sum over the 8 lanes (lane&7) that share one head
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__global__ __launch_bounds__(256) void msda_bwd_m8d32(const float *__restrict__ value,
                                                      const int64_t *__restrict__ shapes,
                                                      const int64_t *__restrict__ lsi,
                                                      const float *__restrict__ loc,
                                                      const float *__restrict__ aw,
                                                      const float *__restrict__ grad_out, int NQ,
                                                      int Lq, int S, int L, int P,
                                                      float *__restrict__ grad_value,
                                                      float *__restrict__ grad_loc,
                                                      float *__restrict__ grad_aw)
{
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int qi = blk * 4 + (threadIdx.x >> 6);
    if (qi >= NQ) return;                       // whole wave leaves together: shuffles below are safe
    const int m = lane >> 3, cg = lane & 7;
    const int b = qi / Lq;
    const long samp = (long)qi * 8 + m;
    const long chan = (long)b * S * 256 + m * 32 + cg * 4;
    const float4 top = *reinterpret_cast<const float4 *>(grad_out + (long)qi * 256 + m * 32 + cg * 4);
    long wp = samp * (long)(L * P), lp = wp * 2;

    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long lvl = chan + (long)((int)lsi[l]) * 256;
        const float *vl = value + lvl;
        float *gl = grad_value + lvl;
        for (int p = 0; p < P; ++p, ++wp, lp += 2) {
            const float weight = aw[wp];
            const float h_im = loc[lp + 1] * (float)H - 0.5f;
            const float w_im = loc[lp] * (float)W - 0.5f;
            float g_w = 0.f, g_h = 0.f, g_a = 0.f;
            // the in-range test depends on (query, head) only: uniform over the 8 lanes of a head
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                const float tx = top.x * weight, ty = top.y * weight, tz = top.z * weight, tw = top.w * weight;
                float4 gh = make_float4(0.f, 0.f, 0.f, 0.f), gw = gh, val = gh;
#define DFX_CORNER(cond, yy, xx, wgt, GH, GW)                                              \
                if (cond) {                                                                \
                    const int o = ((yy) * W + (xx)) * 256;                                 \
                    const float4 v = *reinterpret_cast<const float4 *>(vl + o);            \
                    gh.x += (GH) * v.x; gh.y += (GH) * v.y; gh.z += (GH) * v.z; gh.w += (GH) * v.w; \
                    gw.x += (GW) * v.x; gw.y += (GW) * v.y; gw.z += (GW) * v.z; gw.w += (GW) * v.w; \
                    val.x += (wgt) * v.x; val.y += (wgt) * v.y; val.z += (wgt) * v.z; val.w += (wgt) * v.w; \
                    atomic_add(gl + o, (wgt) * tx); atomic_add(gl + o + 1, (wgt) * ty);    \
                    atomic_add(gl + o + 2, (wgt) * tz); atomic_add(gl + o + 3, (wgt) * tw);\
                }
                DFX_CORNER(h0 >= 0 && w0 >= 0, h0, w0, hh * hw, -hw, -hh)
                DFX_CORNER(h0 >= 0 && w1 <= W - 1, h0, w1, hh * lw, -lw, hh)
                DFX_CORNER(h1 <= H - 1 && w0 >= 0, h1, w0, lh * hw, hw, -lh)
                DFX_CORNER(h1 <= H - 1 && w1 <= W - 1, h1, w1, lh * lw, lw, lh)
#undef DFX_CORNER
                g_a = top.x * val.x + top.y * val.y + top.z * val.z + top.w * val.w;
                g_w = (float)W * (gw.x * tx + gw.y * ty + gw.z * tz + gw.w * tw);
                g_h = (float)H * (gh.x * tx + gh.y * ty + gh.z * tz + gh.w * tw);
            }
            g_w = head_sum(g_w);
            g_h = head_sum(g_h);
            g_a = head_sum(g_a);
            if (cg == 0) {
                atomic_add(grad_loc + lp, g_w);
                atomic_add(grad_loc + lp + 1, g_h);
                atomic_add(grad_aw + wp, g_a);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_generic(const T *__restrict__ value,
                                                        const int64_t *__restrict__ shapes,
                                                        const int64_t *__restrict__ lsi,
                                                        const T *__restrict__ loc,
                                                        const T *__restrict__ aw,
                                                        const T *__restrict__ grad_out, long total,
                                                        int S, int M, int D, int L, int Lq, int P,
                                                        T *__restrict__ grad_value,
                                                        T *__restrict__ grad_loc,
                                                        T *__restrict__ grad_aw)
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
        const long chan = (long)b * S * row + m * D + c;
        const T top = grad_out[idx];
        long wp = samp * L * P, lp = wp * 2;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const long lvl = chan + (long)((int)lsi[l]) * row;
            const T *vl = value + lvl;
            T *gl = grad_value + lvl;
            for (int p = 0; p < P; ++p, ++wp, lp += 2) {
                const T weight = aw[wp];
                const T h_im = loc[lp + 1] * (T)H - (T)0.5;
                const T w_im = loc[lp] * (T)W - (T)0.5;
                if (!(h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W)) continue;
                const T hf = floor(h_im), wf = floor(w_im);
                const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
                const T lh = h_im - hf, lw = w_im - wf, hh = (T)1 - lh, hw = (T)1 - lw;
                const T tg = top * weight;
                T gh = 0, gw = 0, val = 0;
                if (h0 >= 0 && w0 >= 0) {
                    const long o = (long)(h0 * W + w0) * row;
                    const T v = vl[o];
                    gh -= hw * v; gw -= hh * v; val += hh * hw * v;
                    atomic_add(gl + o, hh * hw * tg);
                }
                if (h0 >= 0 && w1 <= W - 1) {
                    const long o = (long)(h0 * W + w1) * row;
                    const T v = vl[o];
                    gh -= lw * v; gw += hh * v; val += hh * lw * v;
                    atomic_add(gl + o, hh * lw * tg);
                }
                if (h1 <= H - 1 && w0 >= 0) {
                    const long o = (long)(h1 * W + w0) * row;
                    const T v = vl[o];
                    gh += hw * v; gw -= lh * v; val += lh * hw * v;
                    atomic_add(gl + o, lh * hw * tg);
                }
                if (h1 <= H - 1 && w1 <= W - 1) {
                    const long o = (long)(h1 * W + w1) * row;
                    const T v = vl[o];
                    gh += lw * v; gw += lh * v; val += lh * lw * v;
                    atomic_add(gl + o, lh * lw * tg);
                }
                atomic_add(grad_aw + wp, top * val);
                atomic_add(grad_loc + lp, (T)W * gw * tg);
                atomic_add(grad_loc + lp + 1, (T)H * gh * tg);
            }
        }
    }
}

template <typename T>
int backward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw,
                  const T *grad_out, int N, int S, int M, int D, int L, int Lq, int P, T *grad_value,
                  T *grad_loc, T *grad_aw, void *stream, bool fast_ok)
{
    const int rc = dfx::check_dims(value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P);
    if (rc < 0) return rc;
    if (rc == 1 || L == 0 || P == 0) return DFX_OK;
    if (!grad_value || !grad_loc || !grad_aw) return dfx::fail(DFX_EINVAL, "msda backward: null gradient buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long nq = (long)N * Lq;
    if constexpr (sizeof(T) == 4) {
        if (fast_ok && M == 8 && D == 32 && nq < (1L << 29) && dfx::aligned16(value) && dfx::aligned16(grad_out)) {
            hipLaunchKernelGGL(msda_bwd_m8d32, dim3((int)((nq + 3) / 4)), dim3(256), 0, st, value, shapes, lsi,
                               loc, aw, grad_out, (int)nq, Lq, S, L, P, grad_value, grad_loc, grad_aw);
            return dfx::check_launch("msda_bwd_m8d32");
        }
    }
    const long total = nq * M * D;
    hipLaunchKernelGGL((msda_bwd_generic<T>), dim3(dfx::grid_for(total)), dim3(256), 0, st, value, shapes, lsi,
                       loc, aw, grad_out, total, S, M, D, L, Lq, P, grad_value, grad_loc, grad_aw);
    return dfx::check_launch("msda_bwd_generic");
}

}  // namespace

extern "C" int dfx_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                                     const float *loc, const float *aw, const float *grad_out, int N,
                                     int S, int M, int D, int L, int Lq, int P, float *grad_value,
                                     float *grad_loc, float *grad_aw, void *stream)
{
    return backward_impl<float>(value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P, grad_value,
                                grad_loc, grad_aw, stream, true);
}

extern "C" int dfx_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                                     const double *loc, const double *aw, const double *grad_out, int N,
                                     int S, int M, int D, int L, int Lq, int P, double *grad_value,
                                     double *grad_loc, double *grad_aw, void *stream)
{
    return backward_impl<double>(value, shapes, lsi, loc, aw, grad_out, N, S, M, D, L, Lq, P, grad_value,
                                 grad_loc, grad_aw, stream, false);
}
