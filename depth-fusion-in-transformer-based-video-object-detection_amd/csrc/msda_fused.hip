// Fused MSDA front end + sampling (inference) for gfx950.
//
// MSDeformAttn.forward in the reference (/root/reference/models/ops/modules/ms_deform_attn.py:98-114)
// runs, between its Linear layers, a softmax over the L*P logits of every (query, head), the
// location arithmetic  ref + offset / normaliser, and only then the sampling kernel - five
// elementwise launches and two HBM round trips of sampling_locations / attention_weights
// (3 * N*Lq*M*L*P floats written and read back).  This kernel consumes the two Linear outputs
// directly, in the two-phase structure of msda_forward.hip: in phase A each lane owns one
// (query, level, point, head) sample, computes its softmax weight (the L*P logits of its head are
// one or a few 16-byte loads; <= 16 exponentials) and its location, and stages the resulting tap
// in LDS; phase B is the shared 16-byte corner gather (msda_tap.h).
//
// `Lr` is the number of reference-point levels.  Lr == L is the normal module.  Lr > L (with
// L == 1) is the TransVOD temporal decoder: the module there builds a location tensor
// [1,Lq,M,Lr,P,2] and the CUDA op reads it flat as [1,Lq,M,1,P,2] (SURVEY.md 0.6,
// ms_deform_attn_cuda.cu:45-48); row j of that flat view is (query j/(Lr*M), head (j/Lr)%M,
// level j%Lr) of the tensor the module built.  The reference only ever does this with N == 1;
// for N > 1 this kernel applies the same rule to every batch element on its own (N independent
// reference calls), which is what batching the temporal stage over frames needs.
#include "dfx_common.h"
#include "msda_tap.h"

namespace {

using dfx::Tap;
using dfx::xcd_remap;

// Same two-phase structure as msda_fwd_taps (msda_forward.hip): phase A builds one tap per lane -
// here from the raw Linear outputs - and stages it in LDS; phase B is the shared gather.
// LT = levels at compile time (1..4), P = 4, M = 8, D = 32, fp32.
template <int LT, int REFDIM>
__global__ __launch_bounds__(256) void msda_fused_taps(const float *__restrict__ value,
                                                       const int64_t *__restrict__ shapes,
                                                       const int64_t *__restrict__ lsi,
                                                       const float *__restrict__ ref, int Lr,
                                                       const float *__restrict__ off, long off_stride,
                                                       const float *__restrict__ logits, long logit_stride,
                                                       int NQ, int Lq, int S, int iters,
                                                       float *__restrict__ out)
{
    constexpr int QW = 2;
    constexpr int TAPS = QW * LT * 32;
    __shared__ uint4 s_off[4][TAPS];
    __shared__ float4 s_w[4][TAPS];
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
        const int q0 = ((blk * iters + it) * 4 + wave) * QW;
        if (q0 >= NQ) break;
#pragma unroll
        for (int c = 0; c < TAPS / 64; ++c) {
            const int s = c * 64 + lane;                 // slot = ((qq*LT + l)*4 + p)*8 + head
            const int hm = s & 7, p = (s >> 3) & 3, ql = s >> 5;
            const int l = (LT == 1) ? 0 : ql % LT, qq = (LT == 1) ? ql : ql / LT;
            const int qi = q0 + qq;
            Tap t;
            if (qi < NQ) {
                // ---- attention weight: softmax over the L*P logits of (query, head), F.softmax(x,-1)
                const float *lg = logits + (long)qi * logit_stride + hm * (LT * 4);
                float4 e[LT];
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < LT; ++k) {
                    e[k] = *reinterpret_cast<const float4 *>(lg + k * 4);
                    mx = fmaxf(mx, fmaxf(fmaxf(e[k].x, e[k].y), fmaxf(e[k].z, e[k].w)));
                }
                float sum = 0.f, mine = 0.f;
#pragma unroll
                for (int k = 0; k < LT; ++k) {
                    const float ex = expf(e[k].x - mx), ey = expf(e[k].y - mx);
                    const float ez = expf(e[k].z - mx), ew = expf(e[k].w - mx);
                    sum += ex; sum += ey; sum += ez; sum += ew;
                    if (k == l) mine = (p == 0) ? ex : (p == 1) ? ey : (p == 2) ? ez : ew;
                }
                const float a = mine / sum;
                // ---- location: which (query, head, reference level) of the tensor the module builds,
                //      [Lq,M,Lr,P,2] per batch element, flat row (q*M+m)*L+l of the op's view is
                //      (identity when Lr == L; otherwise L == 1 and the op reads the prefix of the
                //      batch element's tensor flat, see the file header)
                int r = l, ms = hm, lo = l;
                long qs = qi;
                if (Lr != LT) {
                    const int b = qi / Lq;
                    const long j = ((long)(qi - b * Lq) * 8 + hm) * LT + l;
                    r = (int)(j % Lr);
                    ms = (int)((j / Lr) & 7);
                    qs = (long)b * Lq + j / ((long)Lr * 8);
                    lo = 0;
                }
                const float *rp = ref + (qs * Lr + r) * REFDIM;
                const float2 o2 = *reinterpret_cast<const float2 *>(off + qs * off_stride + ((ms * LT + lo) * 4 + p) * 2);
                int H = Hs[0], W = Ws[0], R = Rs[0];
#pragma unroll
                for (int k = 1; k < LT; ++k)
                    if (l == k) { H = Hs[k]; W = Ws[k]; R = Rs[k]; }
                float x, y;
                if (REFDIM == 2) {
                    // ref + off / (W_lo, H_lo)   (ms_deform_attn.py:102-107); lo == l unless Lr != L (then L == 1)
                    x = rp[0] + o2.x / (float)W;
                    y = rp[1] + o2.y / (float)H;
                } else {
                    // ref_xy + off / P * ref_wh * 0.5   (ms_deform_attn.py:108-110)
                    const float4 rr = *reinterpret_cast<const float4 *>(rp);
                    x = rr.x + o2.x / 4.f * rr.z * 0.5f;
                    y = rr.y + o2.y / 4.f * rr.w * 0.5f;
                }
                t = dfx::make_tap(x, y, a, H, W, R, hm * 128);
            } else {
                t.off = make_uint4(0u, 0u, 0u, 0u);
                t.w = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            toff[s] = t.off;
            tw[s] = t.w;
        }
        dfx::wave_lds_fence();
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
        dfx::wave_lds_fence();
    }
}

template <int LT>
int launch(int ref_dim, const float *value, const int64_t *shapes, const int64_t *lsi, const float *ref,
           int Lr, const float *off, long off_stride, const float *logits, long logit_stride, int NQ,
           int Lq, int S, float *out, hipStream_t st)
{
    int iters = 1;
    while (iters < 8 && NQ / (8L * iters * 2) >= 2048) iters *= 2;
    const dim3 grid((unsigned)((NQ + 8L * iters - 1) / (8L * iters))), block(256);
    // algorithmic bytes of this launch: value + (offsets, logits) + out, fp32
    const long bytes = 4L * ((long)(NQ / Lq) * S * 256 + 3L * NQ * 8 * LT * 4 + (long)NQ * 256);
    if (ref_dim == 2)
        dfx::launch_timed(bytes, Lq, S, msda_fused_taps<LT, 2>, grid, block, 0, st, value, shapes, lsi, ref, Lr, off,
                          off_stride, logits, logit_stride, NQ, Lq, S, iters, out);
    else
        dfx::launch_timed(bytes, Lq, S, msda_fused_taps<LT, 4>, grid, block, 0, st, value, shapes, lsi, ref, Lr, off,
                          off_stride, logits, logit_stride, NQ, Lq, S, iters, out);
    return dfx::check_launch("msda_fused_taps");
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
    if (nq >= (1L << 28) || (long)S * 1024 >= (1L << 32)) return dfx::fail(DFX_ERANGE, "msda fused: problem too large");
    if (S == 0) {
        if (hipMemsetAsync(out, 0, sizeof(float) * nq * M * D, static_cast<hipStream_t>(stream)) != hipSuccess)
            return dfx::fail(DFX_ELAUNCH, "msda fused: memset failed");
        return DFX_OK;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (L) {
        case 1: return launch<1>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
        case 2: return launch<2>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
        case 3: return launch<3>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
        default: return launch<4>(ref_dim, value, shapes, lsi, ref, Lr, off, off_stride, logits, logit_stride, (int)nq, Lq, S, out, st);
    }
}
