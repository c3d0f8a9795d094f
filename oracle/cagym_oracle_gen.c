/* cagym_oracle_gen.c -- TEST INFRASTRUCTURE (CPU oracle), twin of csrc/cagym_gen.h.
 * Restates train_agents_random_positions (gym_collision_avoidance/envs/test_cases.py:1362-1463) with
 * is_pose_valid (:129-133): four uniform draws per attempt (start x, y, goal x, y), start >= min_sep from the
 * earlier starts, goal >= min_sep from the earlier goals, |goal - start| >= min_travel.  The random numbers are
 * counter-based (splitmix64 finaliser), not numpy's MT19937 stream: agreement with the reference is
 * distributional (tests/golden/scenario_stats.npz), agreement with the device kernel is bit for bit. */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01(uint64_t seed, uint32_t s, uint32_t k) {
    uint64_t h = mix64(seed ^ mix64(((uint64_t)s << 32) | (uint64_t)k));
    return (double)(h >> 11) * 0x1.0p-53;
}
/* np.linalg.norm of a 2-vector as this NumPy/OpenBLAS evaluates it ("BLAS rounding") */
static double norm2(double x, double y) { return sqrt(fma(y, y, x * x)); }

/* params: the fields of cagym_gen_params (include/cagym.h), passed flat.  Returns the number of agents whose
 * rejection loop hit max_tries. */
int cao_generate_scenarios(uint64_t seed, int S, int M, int n_min, int n_max, int ego_policy, int ego_dynamics,
                           int policy_a, int policy_b, int other_dynamics, int max_tries, double p_b, double side,
                           double min_travel, double min_sep, double radius, double pref_speed, double coop,
                           double* agents6, int32_t* policy, int32_t* dyn, int32_t* nagents, double* coop_out) {
    int failed_total = 0;
    for (int s = 0; s < S; s++) {
        uint32_t k = 0;
        int n = n_min + (int)(u01(seed, (uint32_t)s, k++) * (double)(n_max - n_min + 1));
        n = n < n_min ? n_min : (n > n_max ? n_max : n);
        double* A = agents6 + (size_t)s * M * 6;
        for (int i = 0; i < M; i++) {
            double* a = A + i * 6;
            int32_t pol = 0 /* Static */, dy = 0 /* Unicycle */;
            if (i < n) {
                double x0 = 0, y0 = 0, gx = 0, gy = 0;
                int ok = 0;
                for (int tries = 0; tries < max_tries && !ok; tries++) {
                    x0 = -side + (2.0 * side) * u01(seed, (uint32_t)s, k++);
                    y0 = -side + (2.0 * side) * u01(seed, (uint32_t)s, k++);
                    gx = -side + (2.0 * side) * u01(seed, (uint32_t)s, k++);
                    gy = -side + (2.0 * side) * u01(seed, (uint32_t)s, k++);
                    ok = !(norm2(gx - x0, gy - y0) < min_travel);
                    for (int j = 0; j < i && ok; j++) {
                        const double* b = A + j * 6;
                        if (norm2(x0 - b[0], y0 - b[1]) < min_sep) ok = 0;
                        if (norm2(gx - b[2], gy - b[3]) < min_sep) ok = 0;
                    }
                }
                if (!ok) failed_total++;
                a[0] = x0; a[1] = y0; a[2] = gx; a[3] = gy; a[4] = pref_speed; a[5] = radius;
                if (i == 0) {
                    pol = ego_policy;
                    dy = ego_dynamics;
                } else {
                    pol = u01(seed, (uint32_t)s, k++) < p_b ? policy_b : policy_a;
                    dy = other_dynamics;
                }
            } else {
                a[0] = a[1] = a[2] = a[3] = 0.0;
                a[4] = pref_speed;
                a[5] = radius;
            }
            policy[(size_t)s * M + i] = pol;
            dyn[(size_t)s * M + i] = dy;
            coop_out[(size_t)s * M + i] = coop;
        }
        nagents[s] = n;
    }
    return failed_total;
}
