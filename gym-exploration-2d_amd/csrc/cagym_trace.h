// cagym_trace.h -- every diagnostic hook of the step kernels in one place.  In the shipped library all of these macros are
// EMPTY; a diagnostic library (tools/*.py: `build.py --variant <tag> -DCAGYM_...`) turns one family on.  The recorded values
// leave a kernel only through the buffers below and feed no output.
//
//   -DCAGYM_STAMPS     thread 0 of workgroup 0 accumulates s_memtime deltas per phase into g_stamps (cagym_debug_stamps)
//   -DCAGYM_WGTRACE    thread 0 of EVERY workgroup records the 100 MHz s_memrealtime clock at kernel entry, after the prologue,
//                      after each of the first 36 steps and at exit, plus its XCC id (cagym_debug_wgtrace; tools/launch_cost.py,
//                      tools/cfg4_timeline.py use slots 20.. for the sub-phases of a ONE-step launch)
//   -DCAGYM_WAVETRACE  lane 0 of EVERY WAVE of one workgroup stamps s_memtime at the marked points of the first 24 steps:
//                      which wave arrives last at each barrier = the critical chain (tools/wave_trace.py, slow_wg_trace.py)
//   -DCAGYM_PMARK      named comments in the ISA at the phase boundaries (tools/isa_phases.py, tools/isa_budget.py)
#pragma once

// ---- STAMPS ------------------------------------------------------------------------------------------------------------------
#ifdef CAGYM_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_BEGIN() unsigned long long stamp_prev = __builtin_amdgcn_s_memtime()
#define STAMP(i)                                                                  \
    do {                                                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                \
            unsigned long long _t = __builtin_amdgcn_s_memtime();                 \
            g_stamps[i] += _t - stamp_prev;                                       \
            stamp_prev = _t;                                                      \
        }                                                                         \
    } while (0)
// sub-step shares of obstacle_lines_phase3 (slots 0, 9, 10, 11; they are part of phase A's slot as well)
#define OBSTAMP_BEGIN() unsigned long long ob_prev = __builtin_amdgcn_s_memtime()
#define OBSTAMP(i)                                                                \
    do {                                                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                \
            unsigned long long _t = __builtin_amdgcn_s_memtime();                 \
            g_stamps[i] += _t - ob_prev;                                          \
            ob_prev = _t;                                                         \
        }                                                                         \
    } while (0)
// lockstep trip counts of the LP groups (linearProgram2 rounds, linearProgram3 outer / inner rounds): wave 0's maxima
#define LPCOUNT_DECL() int c_lp2 = 0, c_lp3o = 0, c_lp3i = 0
#define LPCOUNT(x) ((x)++)
#define LPCOUNT_OUT(dbg) do { if (dbg) { (dbg)[0] = c_lp2; (dbg)[1] = c_lp3o; (dbg)[2] = c_lp3i; } } while (0)
#define LPCOUNT_DBG_DECL() int dbg[3] = {0, 0, 0}
#define LPCOUNT_DBG() dbg
#define LPCOUNT_FOLD(wave, tid, cnt)                                                                                               \
    do {                                                                                                                           \
        if ((wave) == 0) {                                                                                                         \
            int m0 = dbg[0], m1 = dbg[1], m2 = dbg[2];                                                                             \
            for (int off = 32; off; off >>= 1) {                                                                                   \
                m0 = max(m0, __shfl_xor(m0, off)); m1 = max(m1, __shfl_xor(m1, off)); m2 = max(m2, __shfl_xor(m2, off));           \
            }                                                                                                                      \
            if ((tid) == 0 && blockIdx.x == 0) { g_stamps[12] += m0; g_stamps[13] += m1; g_stamps[14] += m2; g_stamps[15] += (cnt); } \
        }                                                                                                                          \
    } while (0)
#else
#define STAMP_BEGIN() do { } while (0)
#define STAMP(i) do { } while (0)
#define OBSTAMP_BEGIN() do { } while (0)
#define OBSTAMP(i) do { } while (0)
#define LPCOUNT_DECL() do { } while (0)
#define LPCOUNT(x) do { } while (0)
#define LPCOUNT_OUT(dbg) do { } while (0)
#define LPCOUNT_DBG_DECL() do { } while (0)
#define LPCOUNT_DBG() nullptr
#define LPCOUNT_FOLD(wave, tid, cnt) do { } while (0)
#endif

// ---- planner phases (-DCAGYM_STAMPS -DDM_STAMPS, tools/dmcts_phases.py): thread 0 of every workgroup adds its s_memtime ticks per
// phase of a grow to g_stamps[0..9] and counts the grows in g_stamps[15]
#if defined(CAGYM_STAMPS) && defined(DM_STAMPS)
// (the deltas are summed in registers and flushed once per grow: an atomic per stamp made the stamped kernel 3.7 times slower and the
// inner phases' shares meaningless)
#define DMSTAMP_BEGIN() unsigned long long dm_prev = __builtin_amdgcn_s_memtime(), dm_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DMSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); dm_acc[i] += _t - dm_prev; dm_prev = _t; } } while (0)
#define DMSTAMP_COUNT() do { } while (0)
#define DMSTAMP_FLUSH() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 11; _i++) atomicAdd(&g_stamps[_i], dm_acc[_i]); atomicAdd(&g_stamps[15], 1ull); } } while (0)
#else
#define DMSTAMP_BEGIN() do { } while (0)
#define DMSTAMP(i) do { } while (0)
#define DMSTAMP_COUNT() do { } while (0)
#define DMSTAMP_FLUSH() do { } while (0)
#endif

// ---- WGTRACE -----------------------------------------------------------------------------------------------------------------
#ifdef CAGYM_WGTRACE
#define CAGYM_WGTRACE_MAXWG 4096
#define CAGYM_WGTRACE_W 48
__device__ unsigned long long g_wgtrace[CAGYM_WGTRACE_MAXWG * CAGYM_WGTRACE_W];
#define WGTRACE(slot)                                                                                       \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG && (slot) < CAGYM_WGTRACE_W)               \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + (slot)] = __builtin_amdgcn_s_memrealtime();            \
    } while (0)
// a plain value (not a clock) into a slot
#define WGTRACE_VALUE(slot, v)                                                                              \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG)                                           \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + (slot)] = (unsigned long long)(v);                     \
    } while (0)
// the same from lane 0 of wave 1 (the laser chunks of a one-step launch run on waves 1..)
#define WGTRACE_W1(slot)                                                                                    \
    do {                                                                                                    \
        if (threadIdx.x == 64 && blockIdx.x < CAGYM_WGTRACE_MAXWG)                                          \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + (slot)] = __builtin_amdgcn_s_memrealtime();            \
    } while (0)
#define WGTRACE_W1_VALUE(slot, v)                                                                           \
    do {                                                                                                    \
        if (threadIdx.x == 64 && blockIdx.x < CAGYM_WGTRACE_MAXWG)                                          \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + (slot)] = (unsigned long long)(v);                     \
    } while (0)
// busy egos of the step, accumulated above bit 8 of slot 39 (its low bits take the XCC id at exit)
#define WGTRACE_BUSY(cnt)                                                                                   \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG)                                           \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + 39] += (unsigned long long)(cnt) << 8;                 \
    } while (0)
#define WGTRACE_XCC()                                                                                       \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG)                                           \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + 39] |= __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u; /* HW_REG_XCC_ID */ \
    } while (0)
// sub-phases of a ONE-step launch (tools/cfg4_timeline.py): slots 20.. are free when n_steps == 1
#define WGTRACE1(slot) do { if (n_steps == 1) WGTRACE(slot); } while (0)
#else
#define WGTRACE(slot) do { } while (0)
#define WGTRACE_VALUE(slot, v) do { } while (0)
#define WGTRACE_W1(slot) do { } while (0)
#define WGTRACE_W1_VALUE(slot, v) do { } while (0)
#define WGTRACE_BUSY(cnt) do { } while (0)
#define WGTRACE_XCC() do { } while (0)
#define WGTRACE1(slot) do { } while (0)
#endif

// ---- WAVETRACE ---------------------------------------------------------------------------------------------------------------
#ifdef CAGYM_WAVETRACE
#define CAGYM_WT_STEPS 24
#define CAGYM_WT_POINTS 16
__device__ int g_wt_wg = 7;  // the traced workgroup (cagym_debug_wavetrace_select)
__device__ unsigned long long g_wavetrace[CAGYM_WT_STEPS * CAGYM_WT_POINTS * 8];
#define WAVETRACE(t, point)                                                                                           \
    do {                                                                                                              \
        if ((threadIdx.x & 63) == 0 && (int)blockIdx.x == g_wt_wg && (t) < CAGYM_WT_STEPS)                             \
            g_wavetrace[((t) * CAGYM_WT_POINTS + (point)) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
// trace rows of the first LP batch of step t (orca_lp_group stamps points 13, 14 through LPWT), or null
#define LPWT_ROWS(t, base) (((int)blockIdx.x == g_wt_wg && (t) < CAGYM_WT_STEPS && (base) == 0) ? g_wavetrace + (size_t)(t) * CAGYM_WT_POINTS * 8 : nullptr)
#define LPWT(wt, k) do { if ((wt) && (threadIdx.x & 63) == 0) (wt)[(k) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WAVETRACE(t, point) do { } while (0)
#define LPWT_ROWS(t, base) nullptr
#define LPWT(wt, k) do { } while (0)
#endif

// ---- PMARK -------------------------------------------------------------------------------------------------------------------
#ifdef CAGYM_PMARK
#define PMARK(name) asm volatile("; PMARK " name)
#else
#define PMARK(name) do { } while (0)
#endif
