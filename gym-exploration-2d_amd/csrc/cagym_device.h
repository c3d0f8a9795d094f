// cagym_device.h -- device-side state layout and per-agent arithmetic (gfx950 / CDNA4).
//
// One lane = one agent slot; a world (M slots) never straddles a wavefront, so all
// agent<->agent exchange is wave-local: an LDS neighbour tile written and read by the same
// wave (LDS ops of one wave execute in issue order) plus 64-bit ballots for per-world
// reductions.  No MFMA: there is no dense contraction on this path.
//
// Arithmetic follows the reference exactly where it defines masks: pos/heading/time in fp64
// (agent.py:21-25), actions through fp32 (env.py:289), np.dot / np.linalg.norm as
// fma(a1,b1,a0*b0) (see oracle/cagym_oracle.c dot2), everything else un-contracted
// (-ffp-contract=off).  Reference citations: env.py = envs/collision_avoidance_env.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cagym.h"

#define CAGYM_WAVE 64
#define CAGYM_MAPD 300
#define CAGYM_MAPW 10 /* u32 words per raster row (320 bits) */
#define CAGYM_MAXNB_CAP 31 /* most other agents a world can hold (max_agents <= 32) */

static constexpr double kPi = 3.141592653589793;

// Everything the kernels need, passed by value (device pointers + scalars).
struct CagymDev {
    int N, M, S, Kobs, go_mode, collide_static, laserscan;
    int maxnb;  // RVO maxNeighbors (policies/RVOPolicy.py:15,25), 1 .. M - 1
    int ko;     // half-plane rows reserved per ego for obstacle lines: 2 * Kobs when RVO agents live among rectangles, else 0
    double dt;
    double inv_dt;  // 1.0 / dt, divided on the host (IEEE, the double Python's `1 / DT` is): a kernel that divides it itself hoists the
                    // quotient out of the step loop, finds no register for it and reloads it from scratch memory inside S1 every step
    // scenario pool [S, ...]
    const double* sc_agents6;
    const double* sc_heading0;  // may be null
    const int32_t* sc_policy;
    const int32_t* sc_dyn;
    const int32_t* sc_nagents;
    const double* sc_coop;
    const int32_t* sc_nobst;
    const uint32_t* map_bits;  // [S,300,10] or null
    const double* sc_obst;     // [S,Kobs,4] xl, yl, xu, yu or null
    const float4* sc_obst_prep;  // [S,Kobs,4] per rectangle: (xl, yl, xu, yu), unit directions of edges 0-1, 2-3, (convex mask, 0, 0, 0)
    // state [N*M]
    double *px, *py, *vx, *vy, *heading, *heading_ego, *dist_goal, *time_rem, *t;
    double *gx, *gy, *radius, *pref, *speed, *dhead, *aux0, *aux1, *coop;
    float* action;
    uint32_t* status;
    int32_t *step_num, *n_observed;
    // per world [N]
    int32_t *n_agents, *episode, *ep_len;
    float* ep_return;
    float* stat_return;
    int32_t *stat_episodes, *stat_steps, *stat_outcomes;
    float2* lp_vel;       // [N*M] the split step's hand-over (cagym_step_begin -> cagym_step_finish): every RVO ego's new velocity
                          // (Agent::computeNewVelocity's result, or its clipped preferred velocity when no half-plane was violated)
    int32_t* dev_status;  // host-mapped word: CAGYM_DEVERR_* written by a kernel whose bounded wait expired (cagym_spin.h); 0 otherwise
};

struct CagymOut {
    float* obs_oas;
    float* obs_ego;
    float* laserscan;
    float* reward;
    uint8_t* flags;
    uint8_t* game_over;
};

// Per-lane agent record held in registers across a step (and across steps in the rollout).
struct Agent {
    double px, py, vx, vy, h, he, dg, trem, t, gx, gy, r, pref, speed, dh, aux0, aux1, coop;
    double prx, pry;  // ref_prll (agent.py:250-269) of the last update_ego_frame; derived, never stored
    float a0, a1;
    uint32_t st;  // CAGYM_FLAG_* | policy << 8 | dyn << 12
    int step;
};

#define ST_POLICY(st) (((st) >> 8) & 15u)
#define ST_DYN(st) (((st) >> 12) & 15u)

__device__ __forceinline__ double dot2(double a0, double a1, double b0, double b1) { return fma(a1, b1, a0 * b0); }
__device__ __forceinline__ double norm2(double x, double y) { return sqrt(dot2(x, y, x, y)); }
__device__ __forceinline__ double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// util.py:27-32.  Non-finite / absurd inputs (outside any action space) take a closed form so a
// wave can never spin: every loop in this file has an exit every lane reaches.
__device__ __forceinline__ double wrap_angle(double a) {
    if (fabs(a) < 3 * kPi) {
        // at most one turn of either loop of util.py:27-32 (a - 2 pi >= -pi for a >= pi, a + 2 pi < pi for a < -pi): two
        // selects instead of two loops; every angle the step produces (heading + one action) takes this path
        const double lo = a - 2 * kPi, hi = a + 2 * kPi;
        return a >= kPi ? lo : (a < -kPi ? hi : a);
    }
    if (!(fabs(a) <= 64.0 * kPi)) {
        if (!isfinite(a)) return a;
        a = a - 2 * kPi * floor((a + kPi) / (2 * kPi));
    }
    while (a >= kPi) a -= 2 * kPi;
    while (a < -kPi) a += 2 * kPi;
    return a;
}

// Dynamics.update_ego_frame (dynamics/Dynamics.py:14-28) + Agent.get_ref (agent.py:250-269).
// Returns ref_prll in (prx, pry).
__device__ __forceinline__ void update_ego_frame(Agent& A, double& prx, double& pry) {
    double gx = A.gx - A.px, gy = A.gy - A.py;
    double dist = sqrt(gx * gx + gy * gy);
    A.dg = dist;
    prx = gx;
    pry = gy;
    if (dist > 1e-8) {
        prx = gx / dist;
        pry = gy / dist;
    }
    A.prx = prx;
    A.pry = pry;
    double ang = atan2(pry, prx);
    A.he = wrap_angle(A.h - ang);
}

__device__ __forceinline__ void ref_axes(const Agent& A, double& prx, double& pry) {
    double gx = A.gx - A.px, gy = A.gy - A.py;
    double dist = sqrt(gx * gx + gy * gy);
    prx = gx;
    pry = gy;
    if (dist > 1e-8) {
        prx = gx / dist;
        pry = gy / dist;
    }
}

// Agent.__init__ (agent.py:9-109) from scenario row `s` (index into the [S*M] pool).
__device__ __forceinline__ void init_agent(const CagymDev& D, Agent& A, int sidx, int slot, bool active) {
    // every load of the scenario row first (one round trip to HBM: this runs inside the step kernels' S2 when a world restarts;
    // with the policy / dynamics / coefficient loads behind the atan2 below they were a second, dependent one)
    const size_t k = (size_t)sidx * D.M + slot;
    const double* s6 = D.sc_agents6 + k * 6;
    const double coop = D.sc_coop[k];
    const uint32_t pol = (uint32_t)D.sc_policy[k], dyn = (uint32_t)D.sc_dyn[k];
    A.px = s6[0];
    A.py = s6[1];
    A.gx = s6[2];
    A.gy = s6[3];
    A.pref = s6[4];
    A.r = s6[5];
    A.vx = A.vy = 0.0;
    A.speed = 0.0;
    A.dh = 0.0;
    A.aux0 = A.aux1 = 0.0;
    A.a0 = A.a1 = 0.f;
    A.h = D.sc_heading0 ? D.sc_heading0[k] : atan2(A.gy - A.py, A.gx - A.px);
    A.coop = coop;
    A.trem = 3.0 * ((norm2(A.px - A.gx, A.py - A.gy) - 0.75) / A.pref);  // agent.py:59-63
    A.t = 0.0;
    A.step = 0;
    A.st = (pol << 8) | (dyn << 12) | (active ? CAGYM_FLAG_ACTIVE : 0u);
    double prx, pry;
    update_ego_frame(A, prx, pry);
}

__device__ __forceinline__ void load_agent(const CagymDev& D, Agent& A, size_t a) {
    A.px = D.px[a]; A.py = D.py[a]; A.vx = D.vx[a]; A.vy = D.vy[a];
    A.h = D.heading[a]; A.he = D.heading_ego[a]; A.dg = D.dist_goal[a];
    A.trem = D.time_rem[a]; A.t = D.t[a]; A.gx = D.gx[a]; A.gy = D.gy[a];
    A.r = D.radius[a]; A.pref = D.pref[a]; A.speed = D.speed[a]; A.dh = D.dhead[a];
    A.aux0 = D.aux0[a]; A.aux1 = D.aux1[a]; A.coop = D.coop[a];
    A.a0 = D.action[2 * a]; A.a1 = D.action[2 * a + 1];
    A.st = D.status[a]; A.step = D.step_num[a];
    ref_axes(A, A.prx, A.pry);  // same expression as update_ego_frame at the agent's last move (pos/goal unchanged)
}

__device__ __forceinline__ void store_agent(const CagymDev& D, const Agent& A, size_t a, bool constants) {
    D.px[a] = A.px; D.py[a] = A.py; D.vx[a] = A.vx; D.vy[a] = A.vy;
    D.heading[a] = A.h; D.heading_ego[a] = A.he; D.dist_goal[a] = A.dg;
    D.time_rem[a] = A.trem; D.t[a] = A.t; D.speed[a] = A.speed; D.dhead[a] = A.dh;
    D.aux0[a] = A.aux0; D.aux1[a] = A.aux1;
    reinterpret_cast<float2*>(D.action)[a] = make_float2(A.a0, A.a1);
    D.status[a] = A.st; D.step_num[a] = A.step;
    if (constants) {
        D.gx[a] = A.gx; D.gy[a] = A.gy; D.radius[a] = A.r; D.pref[a] = A.pref; D.coop[a] = A.coop;
    }
}

// CARRLPolicy table (policies/CARRLPolicy.py:5-15): np.linspace(-pi/6, pi/6, 11)
__device__ __forceinline__ double carrl_heading(int k) {
    double lo = -(kPi / 6), hi = kPi / 6;
    double step = (hi - lo) / 10.0;
    if (k >= 10) return hi;
    if (k < 0) k = 0;
    return (double)k * step + lo;
}

// A policy that already knows the direction of the move it asks for hands it over: unit vector (c, s) of the angle `ang`.
// take_action then skips the fp64 sincos of the new heading hn when hn = ang + d with |d| < 1e-6 (the action went through
// fp32, so d is ~1e-8): cos(ang + d) = c (1 - d^2/2) - s d, sin(ang + d) = s (1 - d^2/2) + c d, error d^3/6 < 2e-19,
// below the half-ulp of any libm sincos.  The RVO policy does (its action IS atan2 of the fp32 move).
struct HeadingHint {
    double c, s, ang;
    bool valid;
};

// Agent.take_action (agent.py:147-190) + dynamics/*.py.  `act` is the fp32 pair of env.py:289.
// Returns whether the agent moved (false: it was already done).  EGO = false leaves Dynamics.update_ego_frame to the
// caller (the phase-split kernels run it on otherwise idle lanes of the next pair phase; nothing below reads it).
template <bool EGO = true>
__device__ __forceinline__ bool take_action(Agent& A, float act0, float act1, double dt, const HeadingHint* hint = nullptr) {
    A.a0 = act0;  // all_actions row (env.py:289): zeros for an agent that is already done
    A.a1 = act1;
    if (A.st & (CAGYM_FLAG_AT_GOAL | CAGYM_FLAG_RAN_OUT_OF_TIME | CAGYM_FLAG_IN_COLLISION)) {  // agent.py:148-159
        if (A.st & CAGYM_FLAG_AT_GOAL) A.st |= CAGYM_FLAG_WAS_AT_GOAL;
        if (A.st & CAGYM_FLAG_IN_COLLISION) A.st |= CAGYM_FLAG_WAS_IN_COLLISION;
        if (!(A.st & CAGYM_FLAG_AT_GOAL)) A.t += dt;
        A.vx = A.vy = 0.0;
        return false;
    }
    double a0 = (double)act0, a1 = (double)act1;
    double h = A.h, speed, hn;
    const uint32_t dyn = ST_DYN(A.st);
    if (dyn == CAGYM_DYN_UNICYCLE || dyn > CAGYM_DYN_FIRSTORDER) {  // the common model first: no walk through the switch
        speed = a0;
        hn = wrap_angle(a1 + h);
    } else
    switch (dyn) {
        default:
        case CAGYM_DYN_MAXTURNRATE: {
            double tr = clipd(a1 / dt, -3.0, 3.0);
            speed = a0;
            hn = wrap_angle(tr * dt + h);
            break;
        }
        case CAGYM_DYN_MAXACC: {  // aux0 = current_speed, aux1 = current_turning_rate
            double tr = clipd(a1 / dt, -3.0, 3.0);
            double lacc = clipd(2.0 * (a0 - A.aux0), -2.0, 2.0);
            double tacc = clipd(2.0 * (tr - A.aux1), -3.0, 3.0);
            A.aux0 += lacc * dt;
            A.aux0 = clipd(A.aux0, -1.1, 1.1);
            A.aux1 += tacc * dt;
            speed = A.aux0;
            hn = wrap_angle(A.aux1 * dt + h);
            break;
        }
        case CAGYM_DYN_SECONDORDER: {  // aux0 = angular_speed_global_frame
            speed = clipd(norm2(A.vx, A.vy) + a0 * dt, 0.0, 1.0);
            double tr = A.aux0 + a1 * dt;
            A.aux0 = clipd(tr, -3.0, 3.0);
            hn = wrap_angle(A.aux0 * dt + h);
            break;
        }
        case CAGYM_DYN_FIRSTORDER:
            speed = a0;
            hn = wrap_angle(a1 * dt + h);
            break;
    }
    double s, c;
    bool need_trig = true;
    if (speed == 0.0) {
        // the agent turns on the spot: position and velocity are speed * (cos, sin) = 0 whatever the angle (only the sign of
        // the zero could differ from the reference's 0 * cos(hn))
        c = 1.0;
        s = 0.0;
        need_trig = false;
    } else if (hint && hint->valid) {
        double d = hn - hint->ang;  // both in [-pi, pi]: d is ~0 or ~+-2 pi
        d -= (2 * kPi) * rint(d * (1.0 / (2 * kPi)));
        if (fabs(d) < 1e-6) {
            const double q = 1.0 - 0.5 * d * d;
            c = hint->c * q - hint->s * d;
            s = hint->s * q + hint->c * d;
            need_trig = false;
        }
    }
    if (need_trig) sincos(hn, &s, &c);
    double dx = speed * c * dt, dy = speed * s * dt;
    A.px += dx;
    A.py += dy;
    A.vx = speed * c;
    A.vy = speed * s;
    A.speed = speed;
    A.dh = wrap_angle(hn - h);
    A.h = hn;
    if (EGO) {
        double prx, pry;
        update_ego_frame(A, prx, pry);
    }
    double ex = A.px - A.gx, ey = A.py - A.gy;
    if (ex * ex + ey * ey <= 0.75 * 0.75) A.st |= CAGYM_FLAG_AT_GOAL;  // utils/end_conditions.py:3-6
    else A.st &= ~(uint32_t)CAGYM_FLAG_AT_GOAL;
    A.trem -= dt;  // agent.py:184-188
    A.t += dt;
    A.step += 1;
    if (A.trem <= 0.0) A.st |= CAGYM_FLAG_RAN_OUT_OF_TIME;
    return true;
}

// Map.world_coordinates_to_map_indices (Map.py:40-47): gx = floor(150 - y / 0.1), gy = floor(150 + x / 0.1).
// fl(v / 0.1) and fl(10 v) differ by at most 2 ulp, so unless 10 v sits within 1e-7 of an integer the floors agree and the
// fp64 division (30 instructions, 32 of them per laser beam) is a multiplication; the rare near-integer case divides.
__device__ __forceinline__ bool world_to_cell(double x, double y, int& gx, int& gy) {
    const double cell = 0.1, ox = (30 / 2.) / cell;
    double qy = y * 10.0, qx = x * 10.0;
    const bool safe = fabs(qy - rint(qy)) > 1e-7 && fabs(qx - rint(qx)) > 1e-7 && fabs(qy) < 1e6 && fabs(qx) < 1e6;
    if (!safe) {
        qy = y / cell;
        qx = x / cell;
    }
    double fx = floor(ox - qy), fy = floor(ox + qx);
    // clamp before the int conversion (out-of-range doubles): far outside the map either way
    fx = fx < -1e6 ? -1e6 : (fx > 1e6 ? 1e6 : fx);
    fy = fy < -1e6 ? -1e6 : (fy > 1e6 ? 1e6 : fy);
    gx = (int)fx;
    gy = (int)fy;
    return gx >= 0 && gy >= 0 && gx < CAGYM_MAPD && gy < CAGYM_MAPD;
}

__device__ __forceinline__ bool map_bit(const uint32_t* map, int gx, int gy) {
    return (map[gx * CAGYM_MAPW + (gy >> 5)] >> (gy & 31)) & 1u;
}

// wall test of _check_for_collisions (env.py:656-666) with the disk mask of Map.py:67-71: any occupied cell (y, x) with
// (x - pj)^2 + (y - pi)^2 < (radius / 0.1)^2.  Row by row: the cells of a row that lie in the disk are the span |x - pj| <= w,
// w the largest integer with w^2 < r2 - dy^2 (exact in fp64), tested against the bit-packed raster a 32-bit word at a time
// instead of cell by cell (13 rows x 1-2 words for a 0.5 m agent instead of 169 cells).
__device__ __forceinline__ bool wall_collision(const uint32_t* map, double px, double py, double radius) {
    int pi, pj;
    if (!world_to_cell(px, py, pi, pj)) return false;
    double rr = radius / 0.1, r2 = rr * rr;
    int R = (int)ceil(rr) + 1;
    if (R > 64) R = 64;
    if (R <= 8) {
        // the usual case (radius <= 0.7 m).  The half-width w(|dy|) = largest integer with w^2 + dy^2 < r2 (capped at R, the
        // reference's window) only shrinks as |dy| grows: one decrementing sweep gives all of them without a square root;
        // then the <= 17 rows x <= 2 words are requested together (the raster is L2-resident: the latency is what costs).
        int wd[9];
        int w = R;
#pragma unroll
        for (int dy = 0; dy <= 8; dy++) {
            if (dy <= R) {
                while (w >= 0 && !((double)w * (double)w + (double)dy * (double)dy < r2)) w--;
                wd[dy] = w;
            } else {
                wd[dy] = -1;
            }
        }
        uint32_t acc = 0u;
#pragma unroll
        for (int k = -8; k <= 8; k++) {
            const int ww = wd[k < 0 ? -k : k], y = pi + k;
            const bool ok = ww >= 0 && y >= 0 && y < CAGYM_MAPD;
            int x0 = pj - ww, x1 = pj + ww;
            x0 = x0 < 0 ? 0 : x0;
            x1 = x1 >= CAGYM_MAPD ? CAGYM_MAPD - 1 : x1;
            const int w0 = x0 >> 5, w1 = x1 >> 5;  // the span is <= 17 cells: one or two words
            const uint32_t* row = map + (ok ? y : 0) * CAGYM_MAPW;
            const uint32_t a = ok ? row[w0] : 0u, b = (ok && w1 != w0) ? row[w1] : 0u;
            const uint32_t below = (1u << (x0 & 31)) - 1u;                                        // bits under x0
            const uint32_t upto = (x1 & 31) == 31 ? 0xffffffffu : ((1u << ((x1 & 31) + 1)) - 1u);  // bits up to x1
            acc |= w1 == w0 ? (a & upto & ~below) : ((a & ~below) | (b & upto));
        }
        return acc != 0u;
    }
    bool hit = false;
    for (int y = pi - R; y <= pi + R; y++) {
        if (y < 0 || y >= CAGYM_MAPD) continue;
        const double dy = (double)(y - pi);
        const double rem = r2 - dy * dy;  // dx * dx + dy * dy < r2  <=>  dx * dx < rem (the sum is exact: small integers)
        if (!(rem > 0.0)) continue;
        int w = (int)ceil(sqrt(rem)) - 1;
        while ((double)(w + 1) * (double)(w + 1) + dy * dy < r2) w++;
        while (w >= 0 && !((double)w * (double)w + dy * dy < r2)) w--;
        if (w < 0) continue;
        if (w > R) w = R;  // the reference's window is [pj - R, pj + R]
        int x0 = pj - w, x1 = pj + w;
        if (x0 < 0) x0 = 0;
        if (x1 >= CAGYM_MAPD) x1 = CAGYM_MAPD - 1;
        const uint32_t* row = map + y * CAGYM_MAPW;
        for (int wd = x0 >> 5; wd <= (x1 >> 5); wd++) {
            const int lo = wd == (x0 >> 5) ? (x0 & 31) : 0, hi = wd == (x1 >> 5) ? (x1 & 31) : 31;
            const uint32_t m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            hit |= (row[wd] & m) != 0u;
        }
    }
    return hit;
}

// The same test split for the step kernels (wall_prep3 / wall_rows3): the per-agent part - cell, window, the half-width of every row
// distance |dy| <= 8 (4 bits each, 15 = no cell of that row in the disk) - is computed once per agent, the per-row part (one or
// two raster words against the span) on a lane per (agent, row).  wall_collision == OR over the 17 rows.  flags bit 0: the
// agent's cell is in the map and its window fits (R <= 8); bit 1: too large for the window (radius > 0.7 m): tested whole.
struct WallPrep {
    int cell;              // pi | pj << 16
    int flags;
    unsigned long long w;  // half-widths of |dy| = 0 .. 8
};
__device__ __forceinline__ WallPrep wall_prep(double px, double py, double radius) {
    int pi, pj;
    const bool in = world_to_cell(px, py, pi, pj);
    const double rr = radius / 0.1, r2 = rr * rr;
    const int R = (int)ceil(rr) + 1;
    WallPrep o;
    o.cell = in ? (pi | (pj << 16)) : 0;
    o.flags = (in && R <= 8 ? 1 : 0) | (in && R > 8 ? 2 : 0);
    unsigned long long packed = 0ull;
    int w = R > 8 ? -1 : R;
    // w^2 + dy^2 < r2 for integers <=> w^2 + dy^2 <= ceil(r2) - 1: the sweep runs on integers (r2 <= 49 here)
    const int lim = R > 8 ? -1 : (int)ceil(r2) - 1;
#pragma unroll
    for (int dy = 0; dy <= 8; dy++) {
        if (dy > R) w = -1;
        while (w >= 0 && w * w + dy * dy > lim) w--;
        packed |= (unsigned long long)(w < 0 ? 15 : w) << (4 * dy);
    }
    o.w = packed;
    return o;
}
__device__ __forceinline__ bool wall_row_hit(const uint32_t* map, int cell, int flags, unsigned long long widths, int k) {
    const int pi = cell & 0xffff, pj = cell >> 16, ady = k < 0 ? -k : k, y = pi + k;
    const int w = (int)((widths >> (4 * ady)) & 15ull);
    const bool ok = (flags & 1) && w != 15 && y >= 0 && y < CAGYM_MAPD;
    int x0 = ok ? pj - w : 0, x1 = ok ? pj + w : 0;
    x0 = x0 < 0 ? 0 : x0;
    x1 = x1 >= CAGYM_MAPD ? CAGYM_MAPD - 1 : x1;
    const int w0 = x0 >> 5, w1 = x1 >> 5;
    const uint32_t* row = map + (ok ? y : 0) * CAGYM_MAPW;
    const uint32_t a = row[w0], b = row[w1];  // unconditional (clamped) loads: no branch around them
    const uint32_t below = (1u << (x0 & 31)) - 1u;
    const uint32_t upto = (x1 & 31) == 31 ? 0xffffffffu : ((1u << ((x1 & 31) + 1)) - 1u);
    return ok && (w1 == w0 ? (a & upto & ~below) : ((a & ~below) | (b & upto))) != 0u;
}
