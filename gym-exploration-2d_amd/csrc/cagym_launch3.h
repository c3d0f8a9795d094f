// cagym_launch3.h -- host-side seam between the C ABI (cagym_api.hip) and the generation-3 kernel translation units.
//
// Every (lanes, compile-time M, worlds per workgroup, OBST) specialisation of k_step3 / k_rollout3 is compiled in its own
// translation unit (cagym_k3_tu.hip, built once per row of CAGYM_K3_SPECS and OBST in {0, 1}) so that the twelve units
// build in parallel and an edit of the kernels re-links in the time of the slowest one (was: one 2.5-minute unit).
// -DCAGYM_MONOLITHIC makes cagym_api.hip include all of them again (diagnostic builds: tools/build_alt.sh, and a plain
// `hipcc -c cagym_api.hip` of the tree).
#pragma once
#include <hip/hip_runtime.h>
#include "cagym_device.h"

// lanes per workgroup, compile-time M (0 = run-time M), worlds per workgroup (0 = 64 / M worlds, LDS stride 64)
#define CAGYM_K3_SPECS(X) X(256, 10, 4) X(256, 10, 5) X(256, 4, 0) X(256, 20, 2) X(256, 0, 0) X(512, 0, 0)

enum { K3_HALF_NONE = 0, K3_HALF_PRE = 1, K3_HALF_POST = 2 };  // the split step's two launches (cagym_split3.h)

struct K3Launch {
    CagymDev D;
    int half = K3_HALF_NONE;
    const float* ext;   // k_step3: external actions (may be null)
    CagymOut out;
    int n_steps;        // k_rollout3
    int any_rvo;
    bool rollout, auto_reset;
    unsigned grid;
    size_t lds;
    hipStream_t stream;
};

#define CAGYM_K3_DECL(NT, MT, WP)                                  \
    void cagym_k3_launch_##NT##_##MT##_##WP##_0(const K3Launch&);  \
    void cagym_k3_launch_##NT##_##MT##_##WP##_1(const K3Launch&);  \
    void cagym_k3_setattr_##NT##_##MT##_##WP##_0(int lds);         \
    void cagym_k3_setattr_##NT##_##MT##_##WP##_1(int lds);
CAGYM_K3_SPECS(CAGYM_K3_DECL)
#undef CAGYM_K3_DECL
