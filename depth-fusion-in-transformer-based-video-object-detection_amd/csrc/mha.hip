// Fused scaled-dot-product attention for the 300-query layers (include/dfx_mha.h) on the gfx950 matrix
// cores, fp32 (v_mfma_f32_32x32x2_f32), online softmax, nothing but the output leaves the CU.
//
// Wave = 32 queries of one (batch element, head); workgroup = WAVES waves sharing the key / value tiles
// of 32 rows that all of them need (LDS, 8.5 KB).  Everything is computed TRANSPOSED so that no matrix
// ever has to change its register layout between the two products:
//
//   S^T[key][query] = sum_d K[key][d] Q[query][d]      A = K tile (LDS), B = Q^T (registers, pre-scaled)
//   O^T[d][query]  += sum_key V[key][d] P^T[key][query]  A = V^T (LDS),   B = P^T = exp(S^T - m) (registers)
//
// In the MFMA accumulator layout lane l owns COLUMN l&31 - here always a query - and 16 of the 32 rows
// (keys of S^T / channels of O^T; lane half h = l>>5 owns rows (r&3)+8(r>>2)+4h of register r).  The
// B operand of an MFMA takes from lane (column, h) the element of inner index 2u+h - or of ANY inner
// index, as long as A pairs it with the same one.  So the 16 MFMAs of the second product are numbered
// by the accumulator register r of S^T: lane (query, h) feeds p[r] as it stands, and A reads V of key
// (r&3)+8(r>>2)+4h for that lane half.  Softmax state (running maximum m, denominator l) is per query
// = per lane pair (l, l^32): one cross-lane exchange per tile for the maximum, one at the end for l;
// rescaling O^T is a per-lane scalar multiply.  Likewise the first product sums d in the order
// lane half 0: d = 0..15, half 1: d = 16..31, so a lane's 16 K operands are 4 ds_read_b128.
//
// Few (batch element, head) pairs - the block of a rank that owns 4 or 8 frames of a clip: 160-320 workgroups, each walking
// up to 78 key tiles one after another - leave the chip half empty and the launch latency-bound.  GROUPS > 1 puts GROUPS wave
// groups on the same 64 queries, group g taking key tiles g, g + GROUPS, ... with its own LDS tiles; the partial (O^T, m, l)
// meet in LDS at the end and group 0 merges them with the usual log-sum-exp rescaling (same result up to the order of the sums).
#include "dfx_common.h"
#include "dfx_mha.h"
#include <stdlib.h>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int D = 32;          // head dimension
constexpr int TK = 32;         // keys per tile
constexpr int KP = 36;         // K tile row pitch (floats): 9 sixteen-byte slots, conflict-free ds_read_b128
constexpr int WAVES = 2;       // 64 queries per workgroup

template <int GROUPS>
__global__ __launch_bounds__(64 * WAVES * GROUPS) void mha_fwd(const float *__restrict__ q, long q_batch, long q_row,
                                                      const float *__restrict__ k, long k_batch, long k_row,
                                                      const float *__restrict__ v, long v_batch, long v_row,
                                                      float *__restrict__ out, long o_batch, long o_row, int Lq,
                                                      int Lk, float scale)
{
    constexpr int GT = 64 * WAVES;                                     // threads of a wave group
    constexpr int TILE = TK * KP + TK * D;                             // floats of a group's K and V tiles
    constexpr int MERGE = (GROUPS - 1) * WAVES * 18 * 64;              // partial results of groups 1.. (re-uses the tiles' space)
    __shared__ __attribute__((aligned(16))) float smem[GROUPS * TILE > MERGE ? GROUPS * TILE : MERGE];
    const int group = __builtin_amdgcn_readfirstlane((int)threadIdx.x / GT);
    float (*const ks)[KP] = reinterpret_cast<float (*)[KP]>(smem + group * TILE);
    float (*const vs)[D] = reinterpret_cast<float (*)[D]>(smem + group * TILE + TK * KP);
    const int tid = threadIdx.x - group * GT, lane = tid & 63, wave = tid >> 6;      // within the group
    const int col = lane & 31, half = lane >> 5;
    const int hd = blockIdx.y, b = blockIdx.z;
    const int qi = (blockIdx.x * WAVES + wave) * 32 + col;            // this lane's query
    // B operand of the first product: Q[query][16*half .. +15], pre-scaled by scale * log2(e): the scores are kept in the log2
    // domain so that the softmax numerators are one v_exp_f32 each (softmax is invariant under the change of base; one multiply
    // per score less - the attention's vector instructions run on the lanes its MFMAs need)
    const float qs = scale * 1.44269504088896340736f;
    float qv[16];
    {
        const float *qp = q + b * q_batch + (long)min(qi, Lq - 1) * q_row + hd * D + half * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 t = *reinterpret_cast<const float4 *>(qp + c * 4);
            qv[c * 4 + 0] = t.x * qs; qv[c * 4 + 1] = t.y * qs; qv[c * 4 + 2] = t.z * qs; qv[c * 4 + 3] = t.w * qs;
        }
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -INFINITY, lsum = 0.f;                                   // lsum: this lane half's part of the denominator
    const float *kb = k + b * k_batch + hd * D, *vb = v + b * v_batch + hd * D;

    const int ntiles = (Lk + TK - 1) / TK, iters = (ntiles + GROUPS - 1) / GROUPS;
    for (int it = 0; it < iters; ++it) {
        const int j0 = (it * GROUPS + group) * TK;                     // (scalar) this group's tile of the round
        __syncthreads();                                               // everyone is done with the previous tile
        if (j0 < Lk) {
#pragma unroll
        for (int u = 0; u < (TK * D / 4) / (64 * WAVES); ++u) {
            const int e = tid + u * 64 * WAVES, r = e >> 3, c = e & 7;
            const int j = min(j0 + r, Lk - 1);
            *reinterpret_cast<float4 *>(&ks[r][c * 4]) = *reinterpret_cast<const float4 *>(kb + (long)j * k_row + c * 4);
            *reinterpret_cast<float4 *>(&vs[r][c * 4]) = *reinterpret_cast<const float4 *>(vb + (long)j * v_row + c * 4);
        }
        }
        __syncthreads();
        if (j0 >= Lk) continue;                                        // (a group without a tile in the last round)
        // ---- S^T = K Q^T: A = K[key = col][16*half + t], B = qv[t] ----
        float ka[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 t = *reinterpret_cast<const float4 *>(&ks[col][half * 16 + c * 4]);
            ka[c * 4 + 0] = t.x; ka[c * 4 + 1] = t.y; ka[c * 4 + 2] = t.z; ka[c * 4 + 3] = t.w;
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], qv[t], s, 0, 0, 0);
        // ---- online softmax over this tile's keys, per query (= lanes l and l^32 together) ----
        if (j0 + TK > Lk) {                                            // (scalar: only the last tile has keys to mask)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (j0 + (r & 3) + 8 * (r >> 2) + 4 * half >= Lk) s[r] = -INFINITY;
        }
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float resc = __builtin_amdgcn_exp2f(m - mn);             // 0 on the first tile
        m = mn;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - mn);                  // p; 2^(-inf) = 0 for padded keys
            psum += s[r];
            o[r] *= resc;
        }
        lsum = lsum * resc + psum;
        // ---- O^T += V^T P^T: MFMA r pairs key (r&3)+8(r>>2) (+4 for lane half 1) with p[r] ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float va = vs[(r & 3) + 8 * (r >> 2) + 4 * half][col];
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s[r], o, 0, 0, 0);
        }
    }
    float l = lsum + __shfl_xor(lsum, 32);
    if (GROUPS > 1) {
        // groups 1.. leave (O^T, m, l) of their keys in LDS; group 0 folds them in: O = sum_g O_g e^(m_g - m), l likewise
        __syncthreads();                                               // the tiles are no longer read
        if (group > 0) {
            float *mine = smem + ((group - 1) * WAVES + wave) * 18 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64] = o[r];
            mine[16 * 64] = m;
            mine[17 * 64] = l;
        }
        __syncthreads();
        if (group > 0) return;
#pragma unroll
        for (int gq = 1; gq < GROUPS; ++gq) {
            const float *p = smem + ((gq - 1) * WAVES + wave) * 18 * 64 + lane;
            const float mg = p[16 * 64], lg = p[17 * 64];
            const float mn = fmaxf(m, mg);                             // m is finite: group 0 owns tile 0
            const float a = __builtin_amdgcn_exp2f(m - mn), bq = __builtin_amdgcn_exp2f(mg - mn);      // bq = 0 for a group that saw no key (m_g = -inf)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = o[r] * a + p[r * 64] * bq;
            l = l * a + lg * bq;
            m = mn;
        }
    }
    if (qi < Lq) {
        const float inv = 1.f / l;
        float *op = out + b * o_batch + (long)qi * o_row + hd * D;
#pragma unroll
        for (int g = 0; g < 4; ++g)                                    // registers 4g..4g+3 = channels 8g + 4*half + 0..3
            *reinterpret_cast<float4 *>(op + 8 * g + 4 * half) =
                make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
    }
}

}  // namespace

extern "C" int dfx_mha_f32(const float *q, long q_batch, long q_row, const float *k, long k_batch, long k_row,
                           const float *v, long v_batch, long v_row, float *out, long o_batch, long o_row, int B,
                           int heads, int Lq, int Lk, float scale, void *stream)
{
    if (B < 0 || heads <= 0 || Lq < 0 || Lk < 0) return dfx::fail(DFX_EINVAL, "mha: bad dimension");
    if ((long)B * Lq == 0) return DFX_OK;
    if (!q || !k || !v || !out) return dfx::fail(DFX_EINVAL, "mha: null pointer");
    if (Lk == 0) return dfx::fail(DFX_EINVAL, "mha: no keys (softmax over an empty set)");
    if (((q_batch | q_row | k_batch | k_row | v_batch | v_row | o_batch | o_row) & 3) || !dfx::aligned16(q) ||
        !dfx::aligned16(k) || !dfx::aligned16(v) || !dfx::aligned16(out))
        return dfx::fail(DFX_EINVAL, "mha: strides must be multiples of 4 floats, buffers 16-byte aligned");
    if (q_row < heads * D || k_row < heads * D || v_row < heads * D || o_row < heads * D)
        return dfx::fail(DFX_EINVAL, "mha: row strides smaller than heads * 32");
    if (B > 65535 || heads > 65535) return dfx::fail(DFX_ERANGE, "mha: grid too large");
    const dim3 grid((unsigned)((Lq + 32 * WAVES - 1) / (32 * WAVES)), (unsigned)heads, (unsigned)B);
    // wave groups over the keys while the launch leaves CUs idle (see the head of the file); DFX_MHA_GROUPS=1/2/4 forces
    const long blocks = (long)grid.x * grid.y * grid.z;
    int groups = Lk >= 128 && blocks <= 400 ? 4 : Lk >= 128 && blocks <= 800 ? 2 : 1;
    if (dfx::tuning().mha_groups) groups = dfx::tuning().mha_groups;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (groups == 4)
        hipLaunchKernelGGL(mha_fwd<4>, grid, dim3(64 * WAVES * 4), 0, st, q, q_batch, q_row, k, k_batch, k_row, v, v_batch, v_row,
                           out, o_batch, o_row, Lq, Lk, scale);
    else if (groups == 2)
        hipLaunchKernelGGL(mha_fwd<2>, grid, dim3(64 * WAVES * 2), 0, st, q, q_batch, q_row, k, k_batch, k_row, v, v_batch, v_row,
                           out, o_batch, o_row, Lq, Lk, scale);
    else
        hipLaunchKernelGGL(mha_fwd<1>, grid, dim3(64 * WAVES), 0, st, q, q_batch, q_row, k, k_batch, k_row, v, v_batch, v_row,
                           out, o_batch, o_row, Lq, Lk, scale);
    return dfx::check_launch("mha_fwd");
}
