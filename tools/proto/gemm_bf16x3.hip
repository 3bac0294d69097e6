// PROTOTYPE, not part of libdfx.so and not used by any product route:
// an fp32 GEMM emulated on the bf16 matrix pipe.  Every fp32 operand is split into three bf16 pieces
// x = x1 + x2 + x3 (8 + 8 + 8 significand bits: exact unless a piece underflows), and
//     C = sum over (i, j) of A_i x B_j^T      9 products (or the 6 with i + j <= 4)
// is accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  tools/proto_bf16x3.py compiles this file on
// its own, times it on the layer4 conv1 shape and compares the error against fp64 with the exact-fp32 MFMA GEMM's.
//
//   C[M,N] = A[M,K] x B[N,K]^T; As / Bs are the split operands [3][rows][K] bf16.  M, N multiples of 128, K of 32.
//   workgroup 256 threads = 2 x 2 waves on a 128 x 128 tile, wave = 2 x 2 MFMA tiles of 32 x 32; K-step 32 through ONE LDS
//   stage ([3][128 rows][32] bf16 per operand, 48 KB: three workgroups per CU), next step's global loads held in registers
//   across the MFMAs; 16-byte chunks XOR-swizzled by (row >> 2) & 3 so that the 16 lanes of a ds_read_b128 group hit 16
//   different 16-byte bank groups.
#include <hip/hip_runtime.h>
#include <stdint.h>

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ unsigned short bf16_rne(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// x [rows * K] fp32 -> y [3][rows * K] bf16
__global__ void split3_kernel(const float *__restrict__ x, unsigned short *__restrict__ y, long n)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4 *>(x + i);
    const float e[4] = {v.x, v.y, v.z, v.w};
    unsigned short p[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned short a = bf16_rne(e[k]);
        const float r1 = e[k] - bf16_to_f32(a);
        const unsigned short b = bf16_rne(r1);
        const float r2 = r1 - bf16_to_f32(b);
        p[0][k] = a; p[1][k] = b; p[2][k] = bf16_rne(r2);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
        *reinterpret_cast<uint2 *>(y + s * n + i) = make_uint2(p[s][0] | ((unsigned)p[s][1] << 16), p[s][2] | ((unsigned)p[s][3] << 16));
}

template <int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const unsigned short *__restrict__ As, const unsigned short *__restrict__ Bs,
                                                          float *__restrict__ C, int M, int N, int K)
{
    constexpr int BM = 128, BN = 128, BK = 32;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * 3 * BM * BK];      // [A|B][split][row][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    // XCD-aware order: every XCD walks one contiguous range, n fastest
    const int nx = N / BN, nblk = gridDim.x;
    const int q = nblk >> 3, rem = nblk & 7, xcd = blockIdx.x & 7;
    const int lin = xcd * q + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const int m0 = (lin / nx) * BM, n0 = (lin % nx) * BN;
    const long sA = (long)M * K, sB = (long)N * K;

    // staging: 6 + 6 sixteen-byte chunks per thread and K-step
    const unsigned short *ga[6], *gb[6];
    int la[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int f = tid + i * 256, s = f >> 9, row = (f & 511) >> 2, c = f & 3;
        ga[i] = As + s * sA + (long)(m0 + row) * K + c * 8;
        gb[i] = Bs + s * sB + (long)(n0 + row) * K + c * 8;
        la[i] = (s * BM + row) * BK + ((c ^ ((row >> 2) & 3)) * 8);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    u32x4 ra[6], rb[6];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ra[i] = *reinterpret_cast<const u32x4 *>(ga[i] + k0);
            rb[i] = *reinterpret_cast<const u32x4 *>(gb[i] + k0);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            *reinterpret_cast<u32x4 *>(lds + la[i]) = ra[i];
            *reinterpret_cast<u32x4 *>(lds + 3 * BM * BK + la[i]) = rb[i];
        }
    };
    const int steps = K / BK;
    load(0);
    store();
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
        if (t + 1 < steps) load((t + 1) * BK);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 a[3][2], b[3][2];
            const int c = kb * 2 + h;
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ar = wm * 64 + i * 32 + r, br = wn * 64 + i * 32 + r;
                    a[s][i] = *reinterpret_cast<const bf16x8 *>(lds + (s * BM + ar) * BK + ((c ^ ((ar >> 2) & 3)) * 8));
                    b[s][i] = *reinterpret_cast<const bf16x8 *>(lds + 3 * BM * BK + (s * BN + br) * BK + ((c ^ ((br >> 2) & 3)) * 8));
                }
            // smallest terms first: (2,2) (2,1) (1,2) | (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
            constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int p = (NPROD == 9 ? 0 : 3); p < 9; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][i], b[PB[p]][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (t + 1 < steps) {
            store();
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[(long)row * N + n0 + wn * 64 + j * 32 + r] = acc[i][j][e];
            }
}

extern "C" int proto_split3(const float *x, unsigned short *y, long n, void *stream)
{
    if (n & 3) return -1;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int proto_gemm_bf16x3(const unsigned short *As, const unsigned short *Bs, float *C, int M, int N, int K, int nprod,
                                 void *stream)
{
    if (M % 128 || N % 128 || K % 32 || (nprod != 9 && nprod != 6)) return -1;
    const dim3 grid((unsigned)((M / 128) * (N / 128)));
    if (nprod == 9) hipLaunchKernelGGL(gemm_bf16x3_kernel<9>, grid, dim3(256), 0, (hipStream_t)stream, As, Bs, C, M, N, K);
    else hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, As, Bs, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
