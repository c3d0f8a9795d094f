// cagym_gen.h -- on-device scenario generation (include/cagym.h: cagym_generate_scenarios).
// Restates train_agents_random_positions (test_cases.py:1362-1463) + is_pose_valid (:129-133); one lane per
// scenario, rejection sampling with a counter-based generator.  The CPU twin is oracle/cagym_oracle_gen.c.
#pragma once
#include "cagym_device.h"

struct GenDev {
    double* agents6;
    int32_t* policy;
    int32_t* dyn;
    int32_t* nagents;
    double* coop;
    int32_t* nobst;
    int S, M;
};

__device__ __forceinline__ uint64_t gen_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// U[0,1) with 53 random bits, draw k of scenario s
__device__ __forceinline__ double gen_u01(uint64_t seed, uint32_t s, uint32_t k) {
    const uint64_t h = gen_mix64(seed ^ gen_mix64(((uint64_t)s << 32) | (uint64_t)k));
    return (double)(h >> 11) * 0x1.0p-53;
}

__global__ void __launch_bounds__(64) k_generate_scenarios(GenDev G, cagym_gen_params P, int32_t* n_failed) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.S) return;
    const int M = G.M;
    uint32_t k = 0;
    int n = P.n_min + (int)(gen_u01(P.seed, s, k++) * (double)(P.n_max - P.n_min + 1));
    n = n < P.n_min ? P.n_min : (n > P.n_max ? P.n_max : n);
    double* A = G.agents6 + (size_t)s * M * 6;
    int failed = 0;
    for (int i = 0; i < M; i++) {
        double* a = A + i * 6;
        int32_t pol = CAGYM_POL_STATIC, dyn = CAGYM_DYN_UNICYCLE;
        if (i < n) {
            double x0 = 0, y0 = 0, gx = 0, gy = 0;
            bool ok = false;
            for (int tries = 0; tries < P.max_tries && !ok; tries++) {  // every lane leaves after max_tries
                x0 = -P.side + (2.0 * P.side) * gen_u01(P.seed, s, k++);  // np.random.uniform(low, high): low + (high-low)*u
                y0 = -P.side + (2.0 * P.side) * gen_u01(P.seed, s, k++);
                gx = -P.side + (2.0 * P.side) * gen_u01(P.seed, s, k++);
                gy = -P.side + (2.0 * P.side) * gen_u01(P.seed, s, k++);
                ok = !(norm2(gx - x0, gy - y0) < P.min_travel);
                for (int j = 0; j < i && ok; j++) {
                    const double* b = A + j * 6;
                    if (norm2(x0 - b[0], y0 - b[1]) < P.min_sep) ok = false;
                    if (norm2(gx - b[2], gy - b[3]) < P.min_sep) ok = false;
                }
            }
            if (!ok) failed++;
            a[0] = x0; a[1] = y0; a[2] = gx; a[3] = gy; a[4] = P.pref_speed; a[5] = P.radius;
            if (i == 0) {
                pol = P.ego_policy;
                dyn = P.ego_dynamics;
            } else {
                pol = gen_u01(P.seed, s, k++) < P.p_b ? P.policy_b : P.policy_a;
                dyn = P.other_dynamics;
            }
        } else {
            a[0] = a[1] = a[2] = a[3] = 0.0;
            a[4] = P.pref_speed;
            a[5] = P.radius;
        }
        G.policy[(size_t)s * M + i] = pol;
        G.dyn[(size_t)s * M + i] = dyn;
        G.coop[(size_t)s * M + i] = P.coop;
    }
    G.nagents[s] = n;
    G.nobst[s] = 0;
    if (failed) atomicAdd(n_failed, failed);
}
