// Fused MSDA front end + sampling (inference) for gfx950.
//
// MSDeformAttn.forward in the reference (/root/reference/models/ops/modules/ms_deform_attn.py:98-114)
// runs, between its Linear layers, a softmax over the L*P logits of every (query, head), the
// location arithmetic  ref + offset / normaliser, and only then the sampling kernel - five
// elementwise launches and two HBM round trips of sampling_locations / attention_weights
// (3 * N*Lq*M*L*P floats written and read back).  This kernel consumes the two Linear outputs
// directly: same wave = query, lane = (head, channel quad) mapping as msda_forward.hip; each lane
// recomputes its head's softmax (L*P <= 16 exponentials) and its 2*L*P location values in
// registers - redundantly across the 8 lanes of a head, which is cheaper than a cross-lane
// exchange at these sizes - and goes straight to the 16-byte corner gathers.
//
// `Lr` is the number of reference-point levels.  Lr == L is the normal module.  Lr > L (with
// L == 1) is the TransVOD temporal decoder: the module there builds a location tensor
// [N,Lq,M,Lr,P,2] and the CUDA op reads it flat as [N,Lq,M,1,P,2] (SURVEY.md 0.6,
// ms_deform_attn_cuda.cu:45-48); row j of that flat view is (query j/(Lr*M), head (j/Lr)%M,
// level j%Lr) of the tensor the module built, which is what `src_row` below reproduces.
#include "dfx_common.h"

namespace {

using dfx::xcd_remap;

struct Tap {
    int o00, o01, o10, o11;
    float w00, w01, w10, w11;
};

__device__ __forceinline__ Tap make_tap(float lx, float ly, float a, int H, int W)
{
    const float h_im = ly * (float)H - 0.5f;
    const float w_im = lx * (float)W - 0.5f;
    const bool inr = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    const float hf = floorf(fminf(fmaxf(h_im, -1.f), (float)H));
    const float wf = floorf(fminf(fmaxf(w_im, -1.f), (float)W));
    const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const float s = inr ? a : 0.f;
    Tap t;
    t.w00 = (h0 >= 0 && w0 >= 0) ? hh * hw * s : 0.f;
    t.w01 = (h0 >= 0 && w1 <= W - 1) ? hh * lw * s : 0.f;
    t.w10 = (h1 <= H - 1 && w0 >= 0) ? lh * hw * s : 0.f;
    t.w11 = (h1 <= H - 1 && w1 <= W - 1) ? lh * lw * s : 0.f;
    const int y0 = min(max(h0, 0), H - 1), y1 = min(max(h1, 0), H - 1);
    const int x0 = min(max(w0, 0), W - 1), x1 = min(max(w1, 0), W - 1);
    t.o00 = (y0 * W + x0) * 256;
    t.o01 = (y0 * W + x1) * 256;
    t.o10 = (y1 * W + x0) * 256;
    t.o11 = (y1 * W + x1) * 256;
    return t;
}

__device__ __forceinline__ void fma4(float4 &acc, float w, const float4 &v)
{
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z);
    acc.w = fmaf(w, v.w, acc.w);
}

// LT = levels at compile time (1..4), P = 4, M = 8, D = 32, fp32.
template <int LT, int REFDIM>
__global__ __launch_bounds__(256) void msda_fused_m8d32p4(const float *__restrict__ value,
                                                          const int64_t *__restrict__ shapes,
                                                          const int64_t *__restrict__ lsi,
                                                          const float *__restrict__ ref, int Lr,
                                                          const float *__restrict__ off, long off_stride,
                                                          const float *__restrict__ logits, long logit_stride,
                                                          int NQ, int Lq, int S, float *__restrict__ out)
{
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int qi = blk * 4 + (threadIdx.x >> 6);
    if (qi >= NQ) return;
    const int m = lane >> 3, cg = lane & 7;
    const int b = qi / Lq;
    const float *vb = value + (long)b * S * 256 + m * 32 + cg * 4;

    // ---- softmax over the L*P logits of (query, head): F.softmax(x, -1) ----
    float4 e[LT];
    const float *lg = logits + (long)qi * logit_stride + m * (LT * 4);
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        e[l] = *reinterpret_cast<const float4 *>(lg + l * 4);
        mx = fmaxf(mx, fmaxf(fmaxf(e[l].x, e[l].y), fmaxf(e[l].z, e[l].w)));
    }
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        e[l].x = expf(e[l].x - mx); e[l].y = expf(e[l].y - mx);
        e[l].z = expf(e[l].z - mx); e[l].w = expf(e[l].w - mx);
        sum += e[l].x; sum += e[l].y; sum += e[l].z; sum += e[l].w;
    }
    const float inv = 1.f / sum;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long i = (long)qi * 8 + m;     // flat (b,q,m)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        if (H <= 0 || W <= 0) continue;
        const float *vl = vb + (long)((int)lsi[l]) * 256;
        // which (query, head, reference level) the flat row i*L+l of the location tensor is
        const long j = i * LT + l;
        const int r = (int)(j % Lr);
        const int ms = (int)((j / Lr) & 7);
        const long qs = j / ((long)Lr * 8);
        const int lo = (Lr == LT) ? r : 0;
        const float *rp = ref + (qs * Lr + r) * REFDIM;
        const float *op = off + qs * off_stride + (ms * LT + lo) * 8;
        const float4 oa = *reinterpret_cast<const float4 *>(op);
        const float4 ob = *reinterpret_cast<const float4 *>(op + 4);
        float x0, y0, x1, y1, x2, y2, x3, y3;
        if (REFDIM == 2) {
            // ref + off / (W_lo, H_lo)   (ms_deform_attn.py:102-107)
            const float nw = (float)shapes[2 * lo + 1], nh = (float)shapes[2 * lo];
            const float rx = rp[0], ry = rp[1];
            x0 = rx + oa.x / nw; y0 = ry + oa.y / nh; x1 = rx + oa.z / nw; y1 = ry + oa.w / nh;
            x2 = rx + ob.x / nw; y2 = ry + ob.y / nh; x3 = rx + ob.z / nw; y3 = ry + ob.w / nh;
        } else {
            // ref_xy + off / P * ref_wh * 0.5   (ms_deform_attn.py:108-110)
            const float4 rr = *reinterpret_cast<const float4 *>(rp);
            x0 = rr.x + oa.x / 4.f * rr.z * 0.5f; y0 = rr.y + oa.y / 4.f * rr.w * 0.5f;
            x1 = rr.x + oa.z / 4.f * rr.z * 0.5f; y1 = rr.y + oa.w / 4.f * rr.w * 0.5f;
            x2 = rr.x + ob.x / 4.f * rr.z * 0.5f; y2 = rr.y + ob.y / 4.f * rr.w * 0.5f;
            x3 = rr.x + ob.z / 4.f * rr.z * 0.5f; y3 = rr.y + ob.w / 4.f * rr.w * 0.5f;
        }
        const Tap t0 = make_tap(x0, y0, e[l].x * inv, H, W);
        const Tap t1 = make_tap(x1, y1, e[l].y * inv, H, W);
        const Tap t2 = make_tap(x2, y2, e[l].z * inv, H, W);
        const Tap t3 = make_tap(x3, y3, e[l].w * inv, H, W);
#define DFX_LD(t, o) (*reinterpret_cast<const float4 *>(vl + t.o))
        const float4 v00 = DFX_LD(t0, o00), v01 = DFX_LD(t0, o01), v02 = DFX_LD(t0, o10), v03 = DFX_LD(t0, o11);
        const float4 v10 = DFX_LD(t1, o00), v11 = DFX_LD(t1, o01), v12 = DFX_LD(t1, o10), v13 = DFX_LD(t1, o11);
        const float4 v20 = DFX_LD(t2, o00), v21 = DFX_LD(t2, o01), v22 = DFX_LD(t2, o10), v23 = DFX_LD(t2, o11);
        const float4 v30 = DFX_LD(t3, o00), v31 = DFX_LD(t3, o01), v32 = DFX_LD(t3, o10), v33 = DFX_LD(t3, o11);
#undef DFX_LD
        fma4(acc, t0.w00, v00); fma4(acc, t0.w01, v01); fma4(acc, t0.w10, v02); fma4(acc, t0.w11, v03);
        fma4(acc, t1.w00, v10); fma4(acc, t1.w01, v11); fma4(acc, t1.w10, v12); fma4(acc, t1.w11, v13);
        fma4(acc, t2.w00, v20); fma4(acc, t2.w01, v21); fma4(acc, t2.w10, v22); fma4(acc, t2.w11, v23);
        fma4(acc, t3.w00, v30); fma4(acc, t3.w01, v31); fma4(acc, t3.w10, v32); fma4(acc, t3.w11, v33);
    }
    *reinterpret_cast<float4 *>(out + (long)qi * 256 + m * 32 + cg * 4) = acc;
}

template <int LT>
int launch(int ref_dim, const float *value, const int64_t *shapes, const int64_t *lsi, const float *ref,
           int Lr, const float *off, long off_stride, const float *logits, long logit_stride, int NQ,
           int Lq, int S, float *out, hipStream_t st)
{
    const dim3 grid((NQ + 3) / 4), block(256);
    if (ref_dim == 2)
        hipLaunchKernelGGL((msda_fused_m8d32p4<LT, 2>), grid, block, 0, st, value, shapes, lsi, ref, Lr, off,
                           off_stride, logits, logit_stride, NQ, Lq, S, out);
    else
        hipLaunchKernelGGL((msda_fused_m8d32p4<LT, 4>), grid, block, 0, st, value, shapes, lsi, ref, Lr, off,
                           off_stride, logits, logit_stride, NQ, Lq, S, out);
    return dfx::check_launch("msda_fused_m8d32p4");
}

}  // namespace

extern "C" int dfx_msda_fused_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                                          const float *ref, int ref_dim, int Lr, const float *off,
                                          long off_stride, const float *logits, long logit_stride, int N,
                                          int S, int M, int D, int L, int Lq, int P, float *out,
                                          void *stream)
{
    const int rc = dfx::check_dims(value, shapes, lsi, off, logits, out, N, S, M, D, L, Lq, P);
    if (rc < 0) return rc;
    if (rc == 1) return DFX_OK;
    if (!ref) return dfx::fail(DFX_EINVAL, "msda fused: null reference points");
    if (ref_dim != 2 && ref_dim != 4) return dfx::fail(DFX_EINVAL, "msda fused: ref_dim must be 2 or 4, got %d", ref_dim);
    if (M != 8 || D != 32 || P != 4 || L < 1 || L > 4)
        return dfx::fail(DFX_EINVAL, "msda fused: only M=8, D=32, P=4, 1<=L<=4 is fused (got M=%d D=%d P=%d L=%d); "
                                     "use dfx_msda_forward_f32", M, D, P, L);
    if (Lr != L && !(L == 1 && Lr >= 1))
        return dfx::fail(DFX_EINVAL, "msda fused: Lr (%d) must equal L (%d) unless L == 1", Lr, L);
    if (off_stride < (long)M * L * P * 2 || logit_stride < (long)M * L * P || (off_stride & 3) || (logit_stride & 3))
        return dfx::fail(DFX_EINVAL, "msda fused: bad row strides");
    if (!dfx::aligned16(value) || !dfx::aligned16(out) || !dfx::aligned16(off) || !dfx::aligned16(logits) ||
        (ref_dim == 4 && !dfx::aligned16(ref)))
        return dfx::fail(DFX_EINVAL, "msda fused: buffers must be 16-byte aligned");
    const long nq = (long)N * Lq;
    if (nq >= (1L << 29)) return dfx::fail(DFX_ERANGE, "msda fused: too many queries");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (L) {
        case 1: return launch<1>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
        case 2: return launch<2>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
        case 3: return launch<3>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
        default: return launch<4>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
    }
}
