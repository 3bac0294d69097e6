#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void *lds_ptr;
__global__ void k(const float* g, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n * 4, 0x00020000);
    // lane L loads 16 bytes from element 8*L (so source stride differs from destination stride)
    if (threadIdx.x < 40) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(s), 16, threadIdx.x * 32, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}
int main() {
    float *g, *o; float h[4096], r[1024];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, 4096);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < 300; ++i) { printf("%g ", r[i]); if (i % 16 == 15) printf("\n"); }
    printf("\n");
    return 0;
}
