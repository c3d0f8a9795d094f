// cagym_orca.h -- ORCA half-plane construction + 2-D linear programs on device, fp32 as RVO2.
//
// Replaces policies/RVOPolicy.py:53-117 + the (un-vendored) rvo2 library it drives: restated from
// the published RVO2 v2.0 algorithm (Agent::computeNewVelocity, linearProgram1/2/3; SURVEY.md
// Appendix A).  Only the ego agent's LP is solved.  "Parity unpinned" against the
// real library; bit-parity target is oracle/cagym_oracle.c (same operation order, no FMA).
//
// Lines live in LDS (float4 = point.x, point.y, dir.x, dir.y), slot k of lane l at [k*64 + l]:
// consecutive lanes hit consecutive 16-B slots, so ds_read/write_b128 are conflict-free, and the
// data-dependent indexing of linearProgram1/3 costs no scratch memory.
#pragma once
#include "cagym_device.h"
#include "cagym_trace.h"

#define RVO_EPS 0.00001f

__device__ __forceinline__ float detf(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

__device__ inline bool orca_lp1(const float4* L, int lane, int no, float radius, float ox, float oy, bool dir_opt,
                                float& rx, float& ry, int stride = CAGYM_WAVE) {
    float4 ln = L[no * stride + lane];
    float dot = ln.x * ln.z + ln.y * ln.w;
    float disc = dot * dot + radius * radius - (ln.x * ln.x + ln.y * ln.y);
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float tl = -dot - sq, tr = -dot + sq;
    for (int i = 0; i < no; i++) {
        float4 li = L[i * stride + lane];
        float den = detf(ln.z, ln.w, li.z, li.w);
        float num = detf(li.z, li.w, ln.x - li.x, ln.y - li.y);
        if (fabsf(den) <= RVO_EPS) {
            if (num < 0.0f) return false;
            continue;
        }
        float t = num / den;
        if (den >= 0.0f) tr = tr < t ? tr : t;
        else tl = tl > t ? tl : t;
        if (tl > tr) return false;
    }
    float t;
    if (dir_opt) {
        t = (ox * ln.z + oy * ln.w > 0.0f) ? tr : tl;
    } else {
        t = ln.z * (ox - ln.x) + ln.w * (oy - ln.y);
        if (t < tl) t = tl;
        else if (t > tr) t = tr;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return true;
}

__device__ inline int orca_lp2(const float4* L, int lane, int n, float radius, float ox, float oy, bool dir_opt,
                               float& rx, float& ry, int stride = CAGYM_WAVE) {
    if (dir_opt) {
        rx = ox * radius;
        ry = oy * radius;
    } else if (ox * ox + oy * oy > radius * radius) {
        float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        rx = ox * inv * radius;
        ry = oy * inv * radius;
    } else {
        rx = ox;
        ry = oy;
    }
    for (int i = 0; i < n; i++) {
        float4 li = L[i * stride + lane];
        if (detf(li.z, li.w, li.x - rx, li.y - ry) > 0.0f) {
            float tx = rx, ty = ry;
            if (!orca_lp1(L, lane, i, radius, ox, oy, dir_opt, rx, ry, stride)) {
                rx = tx;
                ry = ty;
                return i;
            }
        }
    }
    return n;
}

__device__ inline void orca_lp3(const float4* L, float4* P, int lane, int n, int begin, float radius, float& rx,
                                float& ry) {
    float distance = 0.0f;
    for (int i = begin; i < n; i++) {
        float4 li = L[i * CAGYM_WAVE + lane];
        if (detf(li.z, li.w, li.x - rx, li.y - ry) > distance) {
            int np = 0;
            for (int j = 0; j < i; j++) {
                float4 lj = L[j * CAGYM_WAVE + lane];
                float4 ln;
                float d = detf(li.z, li.w, lj.z, lj.w);
                if (fabsf(d) <= RVO_EPS) {
                    if (li.z * lj.z + li.w * lj.w > 0.0f) continue;
                    ln.x = 0.5f * (li.x + lj.x);
                    ln.y = 0.5f * (li.y + lj.y);
                } else {
                    float s = detf(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / d;
                    ln.x = li.x + s * li.z;
                    ln.y = li.y + s * li.w;
                }
                float ddx = lj.z - li.z, ddy = lj.w - li.w;
                float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
                ln.z = ddx * inv;
                ln.w = ddy * inv;
                P[np * CAGYM_WAVE + lane] = ln;
                np++;
            }
            float tx = rx, ty = ry;
            if (orca_lp2(P, lane, np, radius, -li.w, li.z, true, rx, ry) < np) {
                rx = tx;
                ry = ty;
            }
            distance = detf(li.z, li.w, li.x - rx, li.y - ry);
        }
    }
}

// Tile of the ego's world in LDS (pre-move state), indexed by wave lane.
struct NbrTile {
    double* px;
    double* py;
    double* vx;
    double* vy;
    double* r;
};

// fp32 view of the ego used by every half-plane of one solve (RVOPolicy.py:65-85)
struct OrcaEgo {
    float px, py, vx, vy, r, pvx, pvy, max_speed, time_step, c;
};

__device__ __forceinline__ OrcaEgo orca_ego(const Agent& A, double dt) {
    OrcaEgo E;
    E.px = (float)A.px; E.py = (float)A.py; E.vx = (float)A.vx; E.vy = (float)A.vy;
    E.r = (float)((1 + 15e-2) * A.r);
    double gx = A.gx - A.px, gy = A.gy - A.py;
    double sc = A.pref / norm2(gx, gy);
    E.pvx = (float)(sc * gx); E.pvy = (float)(sc * gy);
    E.max_speed = (float)A.pref;
    E.time_step = (float)dt;
    E.c = (float)A.coop;
    return E;
}

// Geometry of one ORCA half-plane ego <- other (Agent::computeNewVelocity body): u (the smallest change of
// the relative velocity that leaves the velocity obstacle), the line direction and the squared centre distance.
// Every operation is odd or even under (ego <-> other), and IEEE rounding is sign-symmetric, so the half-plane
// of the reversed pair is exactly (-u, -direction): one evaluation serves both agents of a pair (P1 below).
struct OrcaPair {
    float ux, uy, zx, zy, d2;
};
__device__ __forceinline__ OrcaPair orca_pair(float pex, float pey, float vex, float vey, float re, float time_step,
                                              double opx, double opy, double ovx, double ovy, double orad) {
    const float inv_th = 1.0f / 5.0f;
    float ox = (float)opx, oy = (float)opy;
    float rpx = ox - pex, rpy = oy - pey;
    float rvx = vex - (float)ovx, rvy = vey - (float)ovy;
    float ro = (float)((1 + 15e-2) * orad);
    float d2 = rpx * rpx + rpy * rpy;
    float cr = re + ro, crsq = cr * cr;
    OrcaPair g;
    g.d2 = d2;
    // The three cases of Agent::computeNewVelocity (cut-off circle, legs, already colliding) share one sqrt and one
    // reciprocal: lanes of a wave take different cases all the time, so the work common to them is done once, with
    // selects, and each case keeps exactly its own expressions (same operands, same order: results are unchanged).
    const bool collide = !(d2 > crsq);
    const float k = collide ? 1.0f / time_step : inv_th;  // 1 / timeStep in the colliding case, 1 / timeHorizon otherwise
    const float wx = rvx - k * rpx, wy = rvy - k * rpy;
    const float wlsq = wx * wx + wy * wy;
    const float dp1 = wx * rpx + wy * rpy;
    const bool circle = collide || (dp1 < 0.0f && dp1 * dp1 > crsq * wlsq);
    const float sq = sqrtf(circle ? wlsq : d2 - crsq);  // |w|, or the leg length
    const float inv = 1.0f / (circle ? sq : d2);
    if (circle) {
        const float uwx = wx * inv, uwy = wy * inv;
        g.zx = uwy;
        g.zy = -uwx;
        const float s = cr * k - sq;
        g.ux = s * uwx;
        g.uy = s * uwy;
    } else {
        const float leg = sq;
        if (detf(rpx, rpy, wx, wy) > 0.0f) {
            g.zx = (rpx * leg - rpy * cr) * inv;
            g.zy = (rpx * cr + rpy * leg) * inv;
        } else {
            g.zx = -((rpx * leg + rpy * cr) * inv);
            g.zy = -((-rpx * cr + rpy * leg) * inv);
        }
        const float dp2 = rvx * g.zx + rvy * g.zy;
        g.ux = dp2 * g.zx - rvx;
        g.uy = dp2 * g.zy - rvy;
    }
    return g;
}

// One ORCA half-plane ego <- other; c = the ego's share of the avoidance (collaboration coefficient).
__device__ __forceinline__ float4 orca_line(float pex, float pey, float vex, float vey, float re, float c,
                                            float time_step, double opx, double opy, double ovx, double ovy, double orad) {
    const OrcaPair g = orca_pair(pex, pey, vex, vey, re, time_step, opx, opy, ovx, ovy, orad);
    return make_float4(vex + c * g.ux, vey + c * g.uy, g.zx, g.zy);
}

// Agent::update (fp32) and the fp64 tail of RVOPolicy.find_next_action (RVOPolicy.py:91-106).
__device__ inline void orca_post(const Agent& A, float nvx, float nvy, double dt, double inv_dt, double& out_speed, double& out_dh,
                                 HeadingHint* hint = nullptr) {
    const float time_step = (float)dt;
    float npx = (float)A.px + nvx * time_step, npy = (float)A.py + nvy * time_step;  // Agent::update, fp32
    double dpx = (double)npx - A.px, dpy = (double)npy - A.py;                      // back in Python, fp64
    double ang1 = atan2(dpy, dpx);
    /* (ang1 - 0) % (2*pi), Python float modulo; |ang1| <= pi so fmod(ang1, 2*pi) == ang1 exactly */
    double nh = ang1;
    if (nh < 0) nh += 2 * kPi;
    double dh = wrap_angle(nh - A.h);
    const double dpn = norm2(dpx, dpy);
    double speed = inv_dt * dpn;  // 1 / dt * |dp| (RVOPolicy.py:104), inv_dt = the double 1 / dt
    if (hint) {  // the move's own direction: the unicycle's new heading is this angle up to the fp32 rounding of dh
        // 1 / |dp| by v_rsq_f64 + two Newton steps (relative error ~1e-16: the hint's own budget is d^3/6 ~ 1e-19 + this)
        const double q = dot2(dpx, dpy, dpx, dpy);
        double inv = __builtin_amdgcn_rsq(q);
        inv = inv * (1.5 - 0.5 * q * inv * inv);
        inv = inv * (1.5 - 0.5 * q * inv * inv);
        hint->valid = dpn > 1e-9;
        hint->c = dpx * inv;
        hint->s = dpy * inv;
        hint->ang = ang1;
    }
    if (fabs(dh) > kPi / 6) {
        dh = (dh > 0 ? 1.0 : (dh < 0 ? -1.0 : 0.0)) * (kPi / 6);
        speed = 0.;
    }
    out_speed = speed;
    out_dh = dh;
}

// LP2 (+LP3) on the nn sorted lines of `lane`, then Agent::update and RVOPolicy.py:91-106.
__device__ inline void orca_solve(const float4* L, float4* P, int lane, int nn, const OrcaEgo& E, const Agent& A,
                                  double dt, double& out_speed, double& out_dh, HeadingHint* hint = nullptr) {
    float nvx, nvy;
    int fail = orca_lp2(L, lane, nn, E.max_speed, E.pvx, E.pvy, false, nvx, nvy);
    if (fail < nn) orca_lp3(L, P, lane, nn, fail, E.max_speed, nvx, nvy);
    orca_post(A, nvx, nvy, dt, 1 / dt, out_speed, out_dh, hint);
}

// RVOPolicy.find_next_action for the agent on `lane`; base = first lane of its world, n = agents in
// the world, i = own slot.  L, P: LDS line arrays [maxnb][64].  Returns (speed, delta_heading).
__device__ inline void orca_action(const NbrTile& T, float4* L, float4* P, int lane, int base, int n, int i,
                                   const Agent& A, double dt, int maxnb, double& out_speed, double& out_dh,
                                   HeadingHint* hint = nullptr) {
    const OrcaEgo E = orca_ego(A, dt);
    // neighbour selection: nearest first, ties in index order, at most maxNeighbors (Agent::insertAgentNeighbor).
    // rank by counting == the insertion sort's result; ranks >= maxNeighbors are dropped.
    int nn = (n - 1) < maxnb ? (n - 1) : maxnb;
    for (int j = 0; j < n; j++) {
        if (j == i) continue;
        float ox = (float)T.px[base + j], oy = (float)T.py[base + j];
        float dx = E.px - ox, dy = E.py - oy;
        float dsq = dx * dx + dy * dy;
        int rank = 0;
        for (int l = 0; l < n; l++) {
            if (l == i || l == j) continue;
            float qx = E.px - (float)T.px[base + l], qy = E.py - (float)T.py[base + l];
            float qsq = qx * qx + qy * qy;
            rank += (qsq < dsq) || (qsq == dsq && l < j);
        }
        if (rank >= maxnb) continue;
        L[rank * CAGYM_WAVE + lane] = orca_line(E.px, E.py, E.vx, E.vy, E.r, E.c, E.time_step, T.px[base + j],
                                                T.py[base + j], T.vx[base + j], T.vy[base + j], T.r[base + j]);
    }
    orca_solve(L, P, lane, nn, E, A, dt, out_speed, out_dh, hint);
}

// ---- linearProgram2 + linearProgram3 of one ego on a GW-lane group (GW = 4, 8 or 16) ---------------------------
// Lane j of the group holds half-plane j.  Line i only ever meets lines j < i, so GW lanes serve nn <= GW + 1
// half-planes.  linearProgram1's interval clipping is a min/max/any reduction over the lanes j < i (same
// argument as for linearProgram3 below); the reductions are DPP row operations (no LDS round trip).  All lanes of
// a group run the same control flow; the groups of one wave may diverge from each other.
// One reduction level = ONE instruction: v_max/min_f32 with a DPP source operand (the builtin route costs a DPP
// move plus canonicalising maxes per level).  s_nop 1 covers the VALU-write -> DPP-read hazard, which the
// compiler's hazard recogniser does not see inside inline assembly.  Operands are never NaN here.
#define CAGYM_DPP_OP(name, op, ctrl)                                                            \
    __device__ __forceinline__ float name(float v) {                                           \
        float r;                                                                               \
        asm volatile("s_nop 1\n\t" op " %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf"       \
                     : "=&v"(r) : "v"(v));                                                    \
        return r;                                                                              \
    }
CAGYM_DPP_OP(dpp_max_row_mirror, "v_max_f32_dpp", "row_mirror")
CAGYM_DPP_OP(dpp_max_half_mirror, "v_max_f32_dpp", "row_half_mirror")
CAGYM_DPP_OP(dpp_max_quad2, "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
CAGYM_DPP_OP(dpp_max_quad1, "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
CAGYM_DPP_OP(dpp_min_row_mirror, "v_min_f32_dpp", "row_mirror")
CAGYM_DPP_OP(dpp_min_half_mirror, "v_min_f32_dpp", "row_half_mirror")
CAGYM_DPP_OP(dpp_min_quad2, "v_min_f32_dpp", "quad_perm:[2,3,0,1]")
CAGYM_DPP_OP(dpp_min_quad1, "v_min_f32_dpp", "quad_perm:[1,0,3,2]")
template <int GW>
__device__ __forceinline__ float grp_max(float v) {
    if (GW >= 16) v = dpp_max_row_mirror(v);
    if (GW >= 8) v = dpp_max_half_mirror(v);
    v = dpp_max_quad2(v);
    return dpp_max_quad1(v);
}
template <int GW>
__device__ __forceinline__ float grp_min(float v) {
    if (GW >= 16) v = dpp_min_row_mirror(v);
    if (GW >= 8) v = dpp_min_half_mirror(v);
    v = dpp_min_quad2(v);
    return dpp_min_quad1(v);
}

// linearProgram1 on the group: half-plane `ln` against the lanes' own lines (`m0` taking part when `take0`, the lane's
// second line `m1` when `take1`).  dir_opt as in RVO2.  Returns false when infeasible (result untouched).
__device__ __forceinline__ void orca_clip_by(const float4 ln, const float4 mine, float& ltl, float& ltr) {
    const float den = detf(ln.z, ln.w, mine.z, mine.w);
    const float num = detf(mine.z, mine.w, ln.x - mine.x, ln.y - mine.y);
    if (fabsf(den) <= RVO_EPS) {
        if (num < 0.0f) ltl = INFINITY;  // "parallel and outside": forces tLeft > tRight below, i.e. infeasible
    } else {
        const float t = num / den;
        if (den >= 0.0f) ltr = fminf(ltr, t);
        else ltl = fmaxf(ltl, t);
    }
}
// The chord of a line in the disc of radius `radius` (linearProgram1's first lines): (tLeft, tRight) = -dot -+ sqrt(disc), or
// (+inf, -inf) when the line misses the disc (disc < 0: "return false" - the empty interval fails the tLeft > tRight test instead).
// It depends on the line alone: the lane that owns a line computes it ONCE (round 3) and the group fetches it by lane shuffle in
// every round that projects onto that line, instead of every lane recomputing dot, disc and a correctly rounded sqrt per round.
__device__ __forceinline__ float2 orca_chord(const float4 ln, float radius) {
    const float dot = ln.x * ln.z + ln.y * ln.w;
    const float disc = dot * dot + radius * radius - (ln.x * ln.x + ln.y * ln.y);
    if (disc < 0.0f) return make_float2(INFINITY, -INFINITY);
    const float sq = sqrtf(disc);
    return make_float2(-dot - sq, -dot + sq);
}
template <int GW, bool TWO>
__device__ __forceinline__ bool orca_lp1_group(const float4 ln, float tl, float tr, const float4 m0, bool take0, const float4 m1, bool take1,
                                               float ox, float oy, bool dir_opt, float& rx, float& ry) {
    float ltl = -INFINITY, ltr = INFINITY;
    if (!TWO) {
        if (take0) {  // one candidate per lane: plain selects
            const float den = detf(ln.z, ln.w, m0.z, m0.w);
            const float num = detf(m0.z, m0.w, ln.x - m0.x, ln.y - m0.y);
            if (fabsf(den) <= RVO_EPS) {
                if (num < 0.0f) ltl = INFINITY;  // "parallel and outside": forces tLeft > tRight below, i.e. infeasible
            } else {
                const float t = num / den;
                if (den >= 0.0f) ltr = t;
                else ltl = t;
            }
        }
    } else {
        if (take0) orca_clip_by(ln, m0, ltl, ltr);
        if (take1) orca_clip_by(ln, m1, ltl, ltr);  // lines GW .. 2 GW - 1 (more than GW + 1 neighbours only)
    }
    tl = fmaxf(tl, grp_max<GW>(ltl));
    tr = fminf(tr, grp_min<GW>(ltr));
    if (tl > tr) return false;
    float t;
    if (dir_opt) {
        t = (ox * ln.z + oy * ln.w > 0.0f) ? tr : tl;
    } else {
        t = ln.z * (ox - ln.x) + ln.w * (oy - ln.y);
        if (t < tl) t = tl;
        else if (t > tr) t = tr;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return true;
}

// projection of half-plane lj onto the boundary of li (linearProgram3); returns false for "parallel, same direction"
__device__ __forceinline__ bool orca_project(const float4 li, const float4 lj, float4& pj) {
    const float d = detf(li.z, li.w, lj.z, lj.w);
    bool have = true;
    if (fabsf(d) <= RVO_EPS) {
        if (li.z * lj.z + li.w * lj.w > 0.0f) have = false;
        pj.x = 0.5f * (li.x + lj.x);
        pj.y = 0.5f * (li.y + lj.y);
    } else {
        const float s = detf(lj.z, lj.w, li.x - lj.x, li.y - lj.y) / d;
        pj.x = li.x + s * li.z;
        pj.y = li.y + s * li.w;
    }
    const float ddx = lj.z - li.z, ddy = lj.w - li.w;
    const float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
    pj.z = ddx * inv;
    pj.w = ddy * inv;
    return have;
}

// Agent::computeNewVelocity after the half-planes exist: linearProgram2 on the sorted lines of agent column `a` with
// optimisation velocity (ox, oy), then linearProgram3 from the failing line when infeasible.  Lane j of the group holds
// half-planes j and j + GW.  TWO = false serves nn <= GW + 1 lines: line i only ever meets lines j < i, so half-plane GW
// (lane 0's second line) is only ever TESTED, never clipped against or projected.  TWO = true serves nn <= 2 GW lines
// (maxNeighbors follows Config.MAX_NUM_AGENTS_IN_ENVIRONMENT, RVOPolicy.py:15: 19 lines with 20 agents).  P: private
// scratch of 2 GW projected lines.
// ROWS: compile-time row count of the ego's column (M - 1) or 0: lanes j < ROWS load their first half-plane without
// waiting for nn (a stale row is never looked at: every use is guarded by j < nn).
template <int GW, bool TWO, int ROWS = 0>
__device__ inline void orca_lp_group(const float4* L, float4* P, int a, int j, int nn, float radius, float ox, float oy,
                                     float& rx, float& ry, int stride, int* lp3_flag = nullptr, int* dbg = nullptr,
                                     unsigned long long* wt = nullptr) {
    LPCOUNT_DECL();
    const int gbase = (threadIdx.x & 63) & ~(GW - 1);
    const uint64_t gbits = (1ull << GW) - 1ull;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 l0 = (ROWS >= GW || j < nn) ? L[j * stride + a] : zero4;
    const float4 l1 = j + GW < nn ? L[(j + GW) * stride + a] : zero4;
    if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        rx = ox * inv * radius;
        ry = oy * inv * radius;
    } else {
        rx = ox;
        ry = oy;
    }
    // the chords of the lane's own lines (the second one only where lines GW .. 2 GW - 1 can be projected onto often: with at most GW + 1
    // lines the one line beyond the lanes - the farthest neighbour - computes its chord in the rare round that needs it)
    const float2 c0 = orca_chord(l0, radius), c1 = TWO ? orca_chord(l1, radius) : make_float2(0.f, 0.f);
    asm volatile("" :: "v"(l0.x), "v"(rx));
    LPWT(wt, 13);
    // linearProgram2.  The reference walks the lines in order and projects onto each violated one; between two
    // projections the result does not change, so "the next violated line at or after cur" is one parallel test
    // (lane j tests its half-planes) + a find-first-set.  The groups of a wave then run linearProgram1 in lockstep,
    // each on its own line index, instead of diverging over an unrolled loop on i.
    int fail = nn;
    for (int cur = 0; cur < nn;) {  // cur grows by at least 1 per trip: every lane leaves after <= nn trips
        const bool v0 = j >= cur && j < nn && detf(l0.z, l0.w, l0.x - rx, l0.y - ry) > 0.0f;
        const bool v1 = j + GW >= cur && j + GW < nn && detf(l1.z, l1.w, l1.x - rx, l1.y - ry) > 0.0f;
        const uint32_t m = (uint32_t)((__ballot(v0) >> gbase) & gbits) | ((uint32_t)((__ballot(v1) >> gbase) & gbits) << GW);
        if (!m) break;
        const int i = __ffs((int)m) - 1;
        LPCOUNT(c_lp2);
        const float4 li = L[i * stride + a];
        const int own = gbase + (i & (GW - 1));  // the lane of the group that owns line i (slot i / GW)
        float tl0, tr0;
        if (!TWO && i >= GW) {  // group-uniform
            const float2 ci = orca_chord(li, radius);
            tl0 = ci.x;
            tr0 = ci.y;
        } else {
            tl0 = __shfl((TWO && i >= GW) ? c1.x : c0.x, own, 64);
            tr0 = __shfl((TWO && i >= GW) ? c1.y : c0.y, own, 64);
        }
        if (!orca_lp1_group<GW, TWO>(li, tl0, tr0, l0, j < i, l1, j + GW < i, ox, oy, false, rx, ry)) {
            fail = i;  // result keeps the value it had before this line (tempResult)
            break;
        }
        cur = i + 1;
    }
    // linearProgram3 (rare).  Same scan-and-jump: the next line at or after `cur` that is violated by more than
    // `distance`, then linearProgram2 (directionOpt) over the projected lines, again by scan-and-jump.
    float distance = 0.0f;
    LPWT(wt, 14);
    if (lp3_flag && fail < nn && j == 0) *lp3_flag = 1;  // this workgroup is in a crowd: its step is the long one
    for (int cur = fail; cur < nn;) {
        const bool w0 = j >= cur && j < nn && detf(l0.z, l0.w, l0.x - rx, l0.y - ry) > distance;
        const bool w1 = j + GW >= cur && j + GW < nn && detf(l1.z, l1.w, l1.x - rx, l1.y - ry) > distance;
        const uint32_t wm = (uint32_t)((__ballot(w0) >> gbase) & gbits) | ((uint32_t)((__ballot(w1) >> gbase) & gbits) << GW);
        if (!wm) break;
        const int i = __ffs((int)wm) - 1;
        cur = i + 1;
        LPCOUNT(c_lp3o);
        const float4 li = L[i * stride + a];
        // projected lines of lane j (those of its half-planes that come before i); `have` = it exists
        bool have0 = false, have1 = false;
        float4 p0 = zero4, p1 = zero4;
        float2 pc0 = make_float2(0.f, 0.f), pc1 = make_float2(0.f, 0.f);  // chords of the lane's projected lines
        if (j < i) {
            have0 = orca_project(li, l0, p0);
            if (have0) { P[j] = p0; pc0 = orca_chord(p0, radius); }
        }
        if (TWO && j + GW < i) {
            have1 = orca_project(li, l1, p1);
            if (have1) { P[j + GW] = p1; pc1 = orca_chord(p1, radius); }
        }
        const float px = -li.w, py = li.z;
        const float tx = rx, ty = ry;
        float qx = px * radius, qy = py * radius;  // linearProgram2, directionOpt
        bool failed = false;
        for (int kcur = 0; kcur < i;) {
            const bool u0 = have0 && j >= kcur && detf(p0.z, p0.w, p0.x - qx, p0.y - qy) > 0.0f;
            const bool u1 = TWO && have1 && j + GW >= kcur && detf(p1.z, p1.w, p1.x - qx, p1.y - qy) > 0.0f;
            uint32_t um = (uint32_t)((__ballot(u0) >> gbase) & gbits);
            if (TWO) um |= (uint32_t)((__ballot(u1) >> gbase) & gbits) << GW;
            if (!um) break;
            const int k = __ffs((int)um) - 1;
            kcur = k + 1;
            LPCOUNT(c_lp3i);
            const float4 pk = P[k];  // projected line k (same wave: the LDS write above is ordered before this read)
            const int ownk = gbase + (k & (GW - 1));
            const float tlk = __shfl((TWO && k >= GW) ? pc1.x : pc0.x, ownk, 64), trk = __shfl((TWO && k >= GW) ? pc1.y : pc0.y, ownk, 64);
            if (!orca_lp1_group<GW, TWO>(pk, tlk, trk, p0, j < k && have0, p1, j + GW < k && have1, px, py, true, qx, qy)) {
                failed = true;
                break;
            }
        }
        if (failed) { rx = tx; ry = ty; }
        else { rx = qx; ry = qy; }
        distance = detf(li.z, li.w, li.x - rx, li.y - ry);
    }
    LPCOUNT_OUT(dbg);
}


// ---- static obstacles in the ORCA solve --------------------------------------------------------------------------------
// RVOPolicy.find_next_action hands the world's rectangles to its private simulator (policies/RVOPolicy.py:56-57
// sim.addObstacle, :45 sim.processObstacles at the first call only), timeHorizonObst = RVO_TIME_HORIZON
// (:25-28).  Restated from RVO2 v2.0 (library absent: PARITY UNPINNED; bit-parity target is oracle/cagym_oracle.c, same
// operation order): RVOSimulator::addObstacle, Agent::computeNeighbors / insertObstacleNeighbor and the obstacle half of
// Agent::computeNewVelocity.  Stated deviation (as in the oracle): RVO2's obstacle BSP may split an edge that straddles
// another edge's supporting line and fixes the order of equidistant edges by its traversal; here edges are never split
// and equidistant edges keep (rectangle, edge) index order, which is what RVO2 itself does for a single rectangle.
struct OrcaVertex {
    float x, y, ux, uy;
    bool convex;
};

// counter-clockwise polygon [(xu,yu), (xl,yu), (xl,yl), (xu,yl)] of test_cases.py:2496.  RVOSimulator::addObstacle's unit
// directions (normalize = v * (1 / |v|): not always exactly +-1) and convexity flags are computed once per scenario pool by
// cagym_set_scenarios with the same fp32 expressions and kept beside the rectangle: prep[0] = (xl, yl, xu, yu) as float,
// prep[1] = (u0.x, u0.y, u1.x, u1.y), prep[2] = (u2.x, u2.y, u3.x, u3.y), prep[3].x = convex bits.
__device__ __forceinline__ OrcaVertex orca_rect_vertex(const float4* prep, int k) {
    const float4 r = prep[0];
    const float4 u = prep[1 + (k >> 1)];
    OrcaVertex V;
    V.x = (k == 0 || k == 3) ? r.z : r.x;
    V.y = (k < 2) ? r.w : r.y;
    V.ux = (k & 1) ? u.z : u.x;
    V.uy = (k & 1) ? u.w : u.y;
    V.convex = (__float_as_uint(prep[3].x) >> k) & 1u;
    return V;
}

__device__ __forceinline__ float orca_dist_sq_point_segment(float ax, float ay, float bx, float by, float cx, float cy) {
    const float r = ((cx - ax) * (bx - ax) + (cy - ay) * (by - ay)) / ((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
    if (r < 0.0f) return (cx - ax) * (cx - ax) + (cy - ay) * (cy - ay);
    if (r > 1.0f) return (cx - bx) * (cx - bx) + (cy - by) * (cy - by);
    const float qx = cx - (ax + r * (bx - ax)), qy = cy - (ay + r * (by - ay));
    return qx * qx + qy * qy;
}

// One obstacle edge o1 -> o2 (pv = predecessor of o1) against the nl lines built so far (L[k * stride], k < nl).
// Returns true and fills `out` when the edge contributes a half-plane.
__device__ inline bool orca_obstacle_line(const OrcaVertex& o1, const OrcaVertex& o2, const OrcaVertex& pv, float px, float py,
                                          float vx, float vy, float radius, float inv_tho, const float4* L, int stride, int nl,
                                          float4& out) {
    const float rp1x = o1.x - px, rp1y = o1.y - py, rp2x = o2.x - px, rp2y = o2.y - py;
    for (int j = 0; j < nl; j++) {  // already covered by an earlier obstacle line?
        const float4 lj = L[j * stride];
        if (detf(inv_tho * rp1x - lj.x, inv_tho * rp1y - lj.y, lj.z, lj.w) - inv_tho * radius >= -RVO_EPS &&
            detf(inv_tho * rp2x - lj.x, inv_tho * rp2y - lj.y, lj.z, lj.w) - inv_tho * radius >= -RVO_EPS)
            return false;
    }
    const float dsq1 = rp1x * rp1x + rp1y * rp1y, dsq2 = rp2x * rp2x + rp2y * rp2y;
    const float rsq = radius * radius;
    const float ovx = o2.x - o1.x, ovy = o2.y - o1.y;
    const float s = ((-rp1x) * ovx + (-rp1y) * ovy) / (ovx * ovx + ovy * ovy);
    const float lx = -rp1x - s * ovx, ly = -rp1y - s * ovy;
    const float dsq_line = lx * lx + ly * ly;
    if (s < 0.0f && dsq1 <= rsq) {  // collision with the left vertex; ignored when non-convex
        if (!o1.convex) return false;
        const float nxv = -rp1y, nyv = rp1x, inv = 1.0f / sqrtf(nxv * nxv + nyv * nyv);
        out = make_float4(0.0f, 0.0f, nxv * inv, nyv * inv);
        return true;
    } else if (s > 1.0f && dsq2 <= rsq) {  // collision with the right vertex; the neighbouring edge takes it otherwise
        if (!(o2.convex && detf(rp2x, rp2y, o2.ux, o2.uy) >= 0.0f)) return false;
        const float nxv = -rp2y, nyv = rp2x, inv = 1.0f / sqrtf(nxv * nxv + nyv * nyv);
        out = make_float4(0.0f, 0.0f, nxv * inv, nyv * inv);
        return true;
    } else if (s >= 0.0f && s < 1.0f && dsq_line <= rsq) {  // collision with the segment
        out = make_float4(0.0f, 0.0f, -o1.ux, -o1.uy);
        return true;
    }
    // no collision: legs
    float llx, lly, rlx, rly;
    OrcaVertex a1 = o1, a2 = o2, left_nb = pv;  // obstacle1 / obstacle2 / obstacle1->prevObstacle_ after the substitutions
    bool same = false;
    if (s < 0.0f && dsq_line <= rsq) {  // viewed obliquely: the left vertex defines the velocity obstacle
        if (!o1.convex) return false;
        a2 = o1;
        same = true;
        const float leg1 = sqrtf(dsq1 - rsq);
        llx = (rp1x * leg1 - rp1y * radius) / dsq1; lly = (rp1x * radius + rp1y * leg1) / dsq1;
        rlx = (rp1x * leg1 + rp1y * radius) / dsq1; rly = (-rp1x * radius + rp1y * leg1) / dsq1;
    } else if (s > 1.0f && dsq_line <= rsq) {  // viewed obliquely: the right vertex defines it
        if (!o2.convex) return false;
        a1 = o2;
        same = true;
        left_nb = o1;  // obstacle2->prevObstacle_
        const float leg2 = sqrtf(dsq2 - rsq);
        llx = (rp2x * leg2 - rp2y * radius) / dsq2; lly = (rp2x * radius + rp2y * leg2) / dsq2;
        rlx = (rp2x * leg2 + rp2y * radius) / dsq2; rly = (-rp2x * radius + rp2y * leg2) / dsq2;
    } else {  // usual situation
        if (o1.convex) {
            const float leg1 = sqrtf(dsq1 - rsq);
            llx = (rp1x * leg1 - rp1y * radius) / dsq1; lly = (rp1x * radius + rp1y * leg1) / dsq1;
        } else { llx = -o1.ux; lly = -o1.uy; }
        if (o2.convex) {
            const float leg2 = sqrtf(dsq2 - rsq);
            rlx = (rp2x * leg2 + rp2y * radius) / dsq2; rly = (-rp2x * radius + rp2y * leg2) / dsq2;
        } else { rlx = o1.ux; rly = o1.uy; }
    }
    // legs never point into a neighbouring edge of a convex vertex: the neighbour's cut-off line takes over
    bool left_foreign = false, right_foreign = false;
    if (a1.convex && detf(llx, lly, -left_nb.ux, -left_nb.uy) >= 0.0f) {
        llx = -left_nb.ux; lly = -left_nb.uy;
        left_foreign = true;
    }
    if (a2.convex && detf(rlx, rly, a2.ux, a2.uy) <= 0.0f) {
        rlx = a2.ux; rly = a2.uy;
        right_foreign = true;
    }
    const float lcx = inv_tho * (a1.x - px), lcy = inv_tho * (a1.y - py);  // cut-off centres
    const float rcx = inv_tho * (a2.x - px), rcy = inv_tho * (a2.y - py);
    const float cvx = rcx - lcx, cvy = rcy - lcy;
    const float t = same ? 0.5f : ((vx - lcx) * cvx + (vy - lcy) * cvy) / (cvx * cvx + cvy * cvy);
    const float t_left = (vx - lcx) * llx + (vy - lcy) * lly;
    const float t_right = (vx - rcx) * rlx + (vy - rcy) * rly;
    if ((t < 0.0f && t_left < 0.0f) || (same && t_left < 0.0f && t_right < 0.0f)) {  // left cut-off circle
        const float wx = vx - lcx, wy = vy - lcy, inv = 1.0f / sqrtf(wx * wx + wy * wy);
        const float uwx = wx * inv, uwy = wy * inv;
        out = make_float4(lcx + radius * inv_tho * uwx, lcy + radius * inv_tho * uwy, uwy, -uwx);
        return true;
    } else if (t > 1.0f && t_right < 0.0f) {  // right cut-off circle
        const float wx = vx - rcx, wy = vy - rcy, inv = 1.0f / sqrtf(wx * wx + wy * wy);
        const float uwx = wx * inv, uwy = wy * inv;
        out = make_float4(rcx + radius * inv_tho * uwx, rcy + radius * inv_tho * uwy, uwy, -uwx);
        return true;
    }
    // left leg, right leg or cut-off line, whichever is closest to the velocity
    float dc = INFINITY, dl = INFINITY, dr = INFINITY;
    if (!(t < 0.0f || t > 1.0f || same)) {
        const float qx = vx - (lcx + t * cvx), qy = vy - (lcy + t * cvy);
        dc = qx * qx + qy * qy;
    }
    if (!(t_left < 0.0f)) {
        const float qx = vx - (lcx + t_left * llx), qy = vy - (lcy + t_left * lly);
        dl = qx * qx + qy * qy;
    }
    if (!(t_right < 0.0f)) {
        const float qx = vx - (rcx + t_right * rlx), qy = vy - (rcy + t_right * rly);
        dr = qx * qx + qy * qy;
    }
    float dx, dy, bx, by;
    if (dc <= dl && dc <= dr) {  // cut-off line
        dx = -a1.ux; dy = -a1.uy; bx = lcx; by = lcy;
    } else if (dl <= dr) {  // left leg
        if (left_foreign) return false;
        dx = llx; dy = lly; bx = lcx; by = lcy;
    } else {  // right leg
        if (right_foreign) return false;
        dx = -rlx; dy = -rly; bx = rcx; by = rcy;
    }
    out = make_float4(bx + radius * inv_tho * (-dy), by + radius * inv_tho * dx, dx, dy);
    return true;
}

// Is edge k of the prepared rectangle an obstacle neighbour of the agent (Agent::computeNeighbors: seen from its right side,
// closer than the range)?  Returns its squared distance through dsq.
__device__ __forceinline__ bool orca_edge_is_neighbour(const float4* rect, int k, float px, float py, float range_sq, float& dsq) {
    const OrcaVertex o1 = orca_rect_vertex(rect, k), o2 = orca_rect_vertex(rect, (k + 1) & 3);
    const float left = detf(o1.x - px, o1.y - py, o2.x - o1.x, o2.y - o1.y);  // leftOf(o1, o2, position)
    if (!(left < 0.0f)) return false;
    const float ex = o2.x - o1.x, ey = o2.y - o1.y;
    const float dsq_line = (left * left) / (ex * ex + ey * ey);
    if (!(dsq_line < range_sq)) return false;
    dsq = orca_dist_sq_point_segment(o1.x, o1.y, o2.x, o2.y, px, py);
    return dsq < range_sq;
}

// The half-plane of obstacle edge `id` (= 4 * rectangle + edge) WITHOUT the already-covered test (nl = 0): the line does
// not depend on the lines built before it, so all of an ego's candidate lines can be built side by side.
__device__ __forceinline__ bool orca_obstacle_line_of(const float4* rects, int id, float px, float py, float vx, float vy,
                                                      float radius, float inv_tho, float4& out) {
    const int r = id >> 2, k = id & 3;
    const OrcaVertex o1 = orca_rect_vertex(rects + 4 * r, k), o2 = orca_rect_vertex(rects + 4 * r, (k + 1) & 3),
                     pv = orca_rect_vertex(rects + 4 * r, (k + 3) & 3);
    return orca_obstacle_line(o1, o2, pv, px, py, vx, vy, radius, inv_tho, nullptr, 0, 0, out);
}

// Agent::computeNewVelocity's "already covered" test of edge `id` against one earlier obstacle line
__device__ __forceinline__ bool orca_edge_covered_by(const float4* rects, int id, float px, float py, float radius, float inv_tho,
                                                     const float4 lj) {
    const int r = id >> 2, k = id & 3;
    const OrcaVertex o1 = orca_rect_vertex(rects + 4 * r, k), o2 = orca_rect_vertex(rects + 4 * r, (k + 1) & 3);
    const float rp1x = o1.x - px, rp1y = o1.y - py, rp2x = o2.x - px, rp2y = o2.y - py;
    return detf(inv_tho * rp1x - lj.x, inv_tho * rp1y - lj.y, lj.z, lj.w) - inv_tho * radius >= -RVO_EPS &&
           detf(inv_tho * rp2x - lj.x, inv_tho * rp2y - lj.y, lj.z, lj.w) - inv_tho * radius >= -RVO_EPS;
}

// ---- linearProgram2/3 on a GW-lane group with LPL half-planes per lane and protected obstacle lines ----------------------
// Line q of the solve (q < no: obstacle line q, row q of the ego's column; q >= no: agent line q - no, row ko + q - no)
// lives on lane q % GW, slot q / GW.  linearProgram3 keeps the obstacle lines as they are and projects only the agent
// lines (RVO2: projLines(lines.begin(), lines.begin() + numObstLines)).  n = no + nn <= GW * LPL.  P: LPL * GW entries, or -
// PC ("projected lines compact", the split step's PRE half) - nn <= 2 GW entries: the projected set's obstacle lines are read
// from their rows of L, only the projected agent lines are kept in P (at q - no): same values, half the scratch.
template <int GW, int LPL>
__device__ __forceinline__ uint64_t orca_group_mask(const bool (&v)[LPL], int gbase) {
    const uint64_t gbits = (1ull << GW) - 1ull;
    uint64_t m = 0;
#pragma unroll
    for (int c = 0; c < LPL; c++) m |= ((__ballot(v[c]) >> gbase) & gbits) << (c * GW);
    return m;
}
template <int GW, int LPL>
__device__ __forceinline__ bool orca_lp1_group_n(const float4 ln, float tl, float tr, const float4 (&mine)[LPL], const bool (&take)[LPL],
                                                 float ox, float oy, bool dir_opt, float& rx, float& ry) {
    // (tl, tr): the chord of ln in the disc, from the lane that owns the line (orca_chord); (+inf, -inf) = the line misses the disc
    float ltl = -INFINITY, ltr = INFINITY;
#pragma unroll
    for (int c = 0; c < LPL; c++)
        if (take[c]) orca_clip_by(ln, mine[c], ltl, ltr);
    tl = fmaxf(tl, grp_max<GW>(ltl));
    tr = fminf(tr, grp_min<GW>(ltr));
    if (tl > tr) return false;
    float t;
    if (dir_opt) {
        t = (ox * ln.z + oy * ln.w > 0.0f) ? tr : tl;
    } else {
        t = ln.z * (ox - ln.x) + ln.w * (oy - ln.y);
        if (t < tl) t = tl;
        else if (t > tr) t = tr;
    }
    rx = ln.x + t * ln.z;
    ry = ln.y + t * ln.w;
    return true;
}
// the chord of line q of the solve, held by lane q % GW in slot q / GW, fetched by every lane of the group
template <int GW, int LPL>
__device__ __forceinline__ float2 orca_group_chord(const float2 (&c)[LPL], int q, int gbase) {
    const int slot = q / GW, own = gbase + (q & (GW - 1));
    float sx = c[0].x, sy = c[0].y;
#pragma unroll
    for (int k = 1; k < LPL; k++)
        if (slot == k) { sx = c[k].x; sy = c[k].y; }
    return make_float2(__shfl(sx, own, 64), __shfl(sy, own, 64));
}
template <int GW, int LPL, bool PC = false>
__device__ inline void orca_lp_group_n(const float4* L, float4* P, int a, int j, int no, int nn, int ko, float radius, float ox,
                                       float oy, float& rx, float& ry, int stride, int* lp3_flag = nullptr) {
    const int gbase = (threadIdx.x & 63) & ~(GW - 1);
    const int n = no + nn;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 l[LPL];
#pragma unroll
    for (int c = 0; c < LPL; c++) {
        const int q = j + c * GW;
        // (clamped address, not a conditional load: `cond ? *p : zero` makes the compiler select between an LDS and a stack address -
        // a flat load and 32 bytes of scratch per lane in every OBST kernel; a lane without a line reads row 0 and never uses it)
        l[c] = L[(q < n ? (q < no ? q : ko + q - no) : 0) * stride + a];
    }
    float2 ch[LPL];  // the chords of the lane's own lines, once (every round that projects onto a line fetches its chord by shuffle)
#pragma unroll
    for (int c = 0; c < LPL; c++) ch[c] = orca_chord(l[c], radius);
    if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        rx = ox * inv * radius;
        ry = oy * inv * radius;
    } else {
        rx = ox;
        ry = oy;
    }
    int fail = n;
    for (int cur = 0; cur < n;) {  // cur grows by at least 1 per trip
        bool v[LPL];
#pragma unroll
        for (int c = 0; c < LPL; c++) {
            const int q = j + c * GW;
            v[c] = q >= cur && q < n && detf(l[c].z, l[c].w, l[c].x - rx, l[c].y - ry) > 0.0f;
        }
        const uint64_t m = orca_group_mask<GW, LPL>(v, gbase);
        if (!m) break;
        const int i = __ffsll((unsigned long long)m) - 1;
        const float4 li = L[(i < no ? i : ko + i - no) * stride + a];
        bool take[LPL];
#pragma unroll
        for (int c = 0; c < LPL; c++) take[c] = j + c * GW < i;
        const float2 ci = orca_group_chord<GW, LPL>(ch, i, gbase);
        if (!orca_lp1_group_n<GW, LPL>(li, ci.x, ci.y, l, take, ox, oy, false, rx, ry)) {
            fail = i;
            break;
        }
        cur = i + 1;
    }
    float distance = 0.0f;
    if (lp3_flag && fail < n && j == 0) *lp3_flag = 1;
    for (int cur = fail; cur < n;) {
        bool w[LPL];
#pragma unroll
        for (int c = 0; c < LPL; c++) {
            const int q = j + c * GW;
            w[c] = q >= cur && q < n && detf(l[c].z, l[c].w, l[c].x - rx, l[c].y - ry) > distance;
        }
        const uint64_t wm = orca_group_mask<GW, LPL>(w, gbase);
        if (!wm) break;
        const int i = __ffsll((unsigned long long)wm) - 1;
        cur = i + 1;
        const float4 li = L[(i < no ? i : ko + i - no) * stride + a];
        // projected set: the obstacle lines as they are, then the agent lines before i projected onto line i.  The set is
        // COMPACTED in RVO2 (skipped "parallel, same direction" lines leave no hole); order and membership are all that
        // linearProgram2 depends on, so holes (have = false) are simply never selected here.
        float4 p[LPL];
        float2 pch[LPL];  // chords of the projected set (an obstacle line keeps its own)
        bool have[LPL];
#pragma unroll
        for (int c = 0; c < LPL; c++) {
            const int q = j + c * GW;
            have[c] = false;
            p[c] = zero4;
            pch[c] = ch[c];
            if (q < no) {
                p[c] = l[c];
                have[c] = true;
            } else if (q < i) {
                have[c] = orca_project(li, l[c], p[c]);
                if (have[c]) pch[c] = orca_chord(p[c], radius);
            }
            if (PC) {
                if (have[c] && q >= no) P[q - no] = p[c];
            } else if (have[c]) P[q] = p[c];
        }
        const int np = i > no ? i : no;  // projected lines occupy indices < max(i, no)
        const float px = -li.w, py = li.z;
        const float tx = rx, ty = ry;
        float qx = px * radius, qy = py * radius;  // linearProgram2, directionOpt
        bool failed = false;
        for (int kcur = 0; kcur < np;) {
            bool u[LPL];
#pragma unroll
            for (int c = 0; c < LPL; c++) {
                const int q = j + c * GW;
                u[c] = have[c] && q >= kcur && detf(p[c].z, p[c].w, p[c].x - qx, p[c].y - qy) > 0.0f;
            }
            const uint64_t um = orca_group_mask<GW, LPL>(u, gbase);
            if (!um) break;
            const int k = __ffsll((unsigned long long)um) - 1;
            kcur = k + 1;
            const float4 pk = PC ? *(k < no ? L + (k * stride + a) : P + (k - no)) : P[k];  // (PC: both are LDS addresses)
            bool take[LPL];
#pragma unroll
            for (int c = 0; c < LPL; c++) take[c] = have[c] && j + c * GW < k;
            const float2 ck = orca_group_chord<GW, LPL>(pch, k, gbase);
            if (!orca_lp1_group_n<GW, LPL>(pk, ck.x, ck.y, p, take, px, py, true, qx, qy)) {
                failed = true;
                break;
            }
        }
        if (failed) { rx = tx; ry = ty; }
        else { rx = qx; ry = qy; }
        distance = detf(li.z, li.w, li.x - rx, li.y - ry);
    }
}
