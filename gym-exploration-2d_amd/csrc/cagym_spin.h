// cagym_spin.h -- the ONE bounded poll loop every intra-workgroup wait of csrc/ goes through.
//
// A wave that waits on an LDS counter for the other waves of its workgroup must be able to leave even if the count can never
// arrive: round 3 met a claim loop that the compiler shaped so that the completion count never reached its target, and an
// unbounded `while (atomic_load(..) < target)` then hangs the GPU (and the box).  Here the poll runs at most `limit` times;
// when it gives up the caller records CAGYM_DEVERR_* in the handle's device status word (CagymDev::dev_status: host-mapped
// memory, read by the next entry point of the C ABI, which then fails with CAGYM_E_DEVICE) and goes on with whatever the
// counter says - results of that launch are void, but every wave reaches the end of the grid.
// The loop is a template over (load, pause) so that tests/test_abi.py can compile and exercise exactly this logic on the CPU
// (tests/spin_check.cpp); on the device `load` is a workgroup-scope acquire load and `pause` is s_sleep.
#pragma once

#ifndef CAGYM_SPIN_HD
#ifdef __HIPCC__
#define CAGYM_SPIN_HD __host__ __device__ __forceinline__
#else
#define CAGYM_SPIN_HD inline
#endif
#endif

// polls before a wait is declared lost.  One poll is an LDS load + s_sleep 1 (>= 64 cycles): 2^20 polls are >= 28 ms at 2.4 GHz,
// three orders of magnitude above the longest legitimate wait (one LaserScan pass or one round of linear programs: microseconds)
#define CAGYM_SPIN_LIMIT (1u << 20)

enum { CAGYM_DEVERR_NONE = 0, CAGYM_DEVERR_LP_WAIT = 1, CAGYM_DEVERR_LASER_WAIT = 2 };

// true when load() >= target was seen within `limit` polls (the condition is tested once more after the last pause)
template <typename Load, typename Pause>
CAGYM_SPIN_HD bool cagym_bounded_wait(Load load, int target, unsigned limit, Pause pause) {
    for (unsigned i = 0; i < limit; i++) {
        if (load() >= target) return true;
        pause();
    }
    return load() >= target;
}

#ifdef __HIPCC__
// wait until the workgroup-scope LDS counter *ctr reaches target (acquire); false = gave up
__device__ __forceinline__ bool lds_wait_ge(int* ctr, int target) {
    return cagym_bounded_wait([&]() { return __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }, target,
                              CAGYM_SPIN_LIMIT, []() { __builtin_amdgcn_s_sleep(1); });
}
#endif
