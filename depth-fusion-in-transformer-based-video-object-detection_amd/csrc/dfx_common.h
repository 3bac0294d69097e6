// Shared host/device helpers of the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
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

// ---- tuning / diagnostic knobs ----------------------------------------------------------------
// Every DFX_* environment switch of the launchers (INTEGRATION.md "Diagnostic environment variables"), read ONCE per
// process - a launch costs no getenv() scan and never races a setenv() - and again only on dfx_tuning_reload()
// (dfx.ops.reload_tuning(): what a test or an A/B tool calls after it changed the environment).
struct Tuning {
    int gemm_group = 8;              // DFX_GEMM_GROUP: tile rows per XCD walk
    int gemm_tile = -1;              // DFX_GEMM_TILE: forced tile (0..7), -1 = the rules
    int gemm_rows_max = 4800;        // DFX_GEMM_ROWS_MAX / DFX_GEMM_NO_ROWS (-> 0): row limit of linear_rows_kernel
    bool gemm_old_epilogue = false;  // DFX_GEMM_OLD_EPILOGUE
    bool gemm_no_dma = false;        // DFX_GEMM_NO_DMA
    bool gemm_narrow_epilogue = false;  // DFX_GEMM_NARROW_EPILOGUE
    bool gemm_no_deep = false;       // DFX_GEMM_NO_DEEP
    int mha_groups = 0;              // DFX_MHA_GROUPS: 1 / 2 / 4 forced, 0 = the rule
    int wino_no_phase = -1;          // DFX_WINO_NO_PHASE: 0 / 1 forced, -1 = the rule
    bool wino_no_tail = false;       // DFX_WINO_NO_TAIL
    bool level_not_persistent = false;  // DFX_LEVEL_NOT_PERSISTENT
    bool level_variant_set = false;  // DFX_LEVEL_VARIANT given
    int level_variant = 0;           // DFX_LEVEL_VARIANT: kernel variant of msda_level.hip (A/B)
    void read()
    {
        *this = Tuning();
        auto num = [](const char *n, int dflt) { const char *e = getenv(n); return e ? atoi(e) : dflt; };
        auto flag = [](const char *n) { return getenv(n) != nullptr; };
        gemm_group = num("DFX_GEMM_GROUP", 8);
        if (const char *e = getenv("DFX_GEMM_TILE")) gemm_tile = e[0] >= '0' && e[0] <= '9' ? e[0] - '0' : -1;
        gemm_rows_max = flag("DFX_GEMM_NO_ROWS") ? 0 : num("DFX_GEMM_ROWS_MAX", 4800);
        gemm_old_epilogue = flag("DFX_GEMM_OLD_EPILOGUE");
        gemm_no_dma = flag("DFX_GEMM_NO_DMA");
        gemm_narrow_epilogue = flag("DFX_GEMM_NARROW_EPILOGUE");
        gemm_no_deep = flag("DFX_GEMM_NO_DEEP");
        if (const char *e = getenv("DFX_MHA_GROUPS")) mha_groups = e[0] == '4' ? 4 : e[0] == '2' ? 2 : 1;
        if (const char *e = getenv("DFX_WINO_NO_PHASE")) wino_no_phase = e[0] == '1';
        wino_no_tail = flag("DFX_WINO_NO_TAIL");
        level_not_persistent = flag("DFX_LEVEL_NOT_PERSISTENT");
        level_variant_set = flag("DFX_LEVEL_VARIANT");
        level_variant = num("DFX_LEVEL_VARIANT", 0);
    }
};
inline Tuning &tuning_slot()
{
    static Tuning t = [] { Tuning x; x.read(); return x; }();
    return t;
}
inline const Tuning &tuning() { return tuning_slot(); }

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
