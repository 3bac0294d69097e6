// Shared host/device helpers of the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>
#include <vector>

#include "dfx_msda.h"

namespace dfx {

// ---- error text, one slot per calling thread -------------------------------------------
inline char *err_slot()
{
    static thread_local char buf[256] = {0};
    return buf;
}

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_slot(), 256, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DFX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DFX_OK;
}

// ---- XCD-aware block remap -------------------------------------------------------------
// MI355X deals workgroups round-robin over its 8 XCDs (block b -> XCD b%8), and every XCD
// has a private 4 MiB L2.  Queries that are neighbours in raster order sample neighbouring
// rows of the value map, so we want each XCD to walk ONE contiguous range of queries.
// Bijective for any grid size: XCD x owns q + (x < r) consecutive logical blocks.
// Placement is a speed matter only; results never depend on it.
__device__ __forceinline__ int xcd_remap(int b, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7;
    const int x = b & 7, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

constexpr int kWave = 64;

// ---- optional per-launch timing -----------------------------------------------------------
// Measurement aid for bench.py (include/dfx_msda.h, dfx_profile_*).  While enabled, kernels launched
// through launch_timed() are dispatched with hipExtLaunchKernelGGL, which stamps a (start, stop) event
// pair with the kernel's own begin / end timestamps - the kernel duration proper, without the
// dispatch gap a pair of hipEventRecord calls around the launch would include.  The pairs are kept in
// a process-wide list until dfx_profile_drain() reads and destroys them.
struct ProfileRecord {
    hipEvent_t start, stop;
    long bytes;
    int tag_a, tag_b;
};
struct ProfileState {
    bool enabled = false;
    std::mutex mu;
    std::vector<ProfileRecord> records;
};
inline ProfileState &profile_state()
{
    static ProfileState s;
    return s;
}

template <typename Kernel, typename... Args>
inline void launch_timed(long bytes, int tag_a, int tag_b, Kernel kernel, dim3 grid, dim3 block, size_t lds,
                         hipStream_t st, Args... args)
{
    ProfileState &p = profile_state();
    if (p.enabled) {
        ProfileRecord r{nullptr, nullptr, bytes, tag_a, tag_b};
        if (hipEventCreate(&r.start) == hipSuccess && hipEventCreate(&r.stop) == hipSuccess) {
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, r.start, r.stop, 0, args...);
            std::lock_guard<std::mutex> lock(p.mu);
            p.records.push_back(r);
            return;
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

// ---- argument checks shared by every MSDA entry point ------------------------------------
// returns <0: error code, 1: empty problem (nothing to launch), 0: go
inline int check_dims(const void *value, const void *shapes, const void *lsi, const void *loc,
                      const void *aw, const void *out, int N, int S, int M, int D, int L, int Lq, int P)
{
    if (N < 0 || S < 0 || M <= 0 || D <= 0 || L < 0 || Lq < 0 || P < 0)
        return fail(DFX_EINVAL, "msda: negative or zero dimension (N=%d S=%d M=%d D=%d L=%d Lq=%d P=%d)",
                    N, S, M, D, L, Lq, P);
    if ((long)N * Lq * M * D == 0) return 1;
    if (!value || !out || (L > 0 && P > 0 && (!shapes || !lsi || !loc || !aw)))
        return fail(DFX_EINVAL, "msda: null pointer");
    // offsets inside one batch element's value slab are 32-bit; everything else is 64-bit
    if ((long)S * M * D >= (1L << 31))
        return fail(DFX_ERANGE, "msda: value slab of one batch element exceeds 2^31 elements");
    return 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int grid_for(long total, int block = 256, int cap = 16384)
{
    const long g = (total + block - 1) / block;
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace dfx
