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

#define RVO_EPS 0.00001f

__device__ __forceinline__ float detf(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

__device__ inline bool orca_lp1(const float4* L, int lane, int no, float radius, float ox, float oy, bool dir_opt,
                                float& rx, float& ry) {
    float4 ln = L[no * CAGYM_WAVE + lane];
    float dot = ln.x * ln.z + ln.y * ln.w;
    float disc = dot * dot + radius * radius - (ln.x * ln.x + ln.y * ln.y);
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float tl = -dot - sq, tr = -dot + sq;
    for (int i = 0; i < no; i++) {
        float4 li = L[i * CAGYM_WAVE + lane];
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
                               float& rx, float& ry) {
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
        float4 li = L[i * CAGYM_WAVE + lane];
        if (detf(li.z, li.w, li.x - rx, li.y - ry) > 0.0f) {
            float tx = rx, ty = ry;
            if (!orca_lp1(L, lane, i, radius, ox, oy, dir_opt, rx, ry)) {
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

// RVOPolicy.find_next_action for the agent on `lane`; base = first lane of its world, n = agents in
// the world, i = own slot.  L, P: LDS line arrays [CAGYM_MAXNB][64].  Returns (speed, delta_heading).
__device__ inline void orca_action(const NbrTile& T, float4* L, float4* P, int lane, int base, int n, int i,
                                   const Agent& A, double dt, double& out_speed, double& out_dh) {
    const float pex = (float)A.px, pey = (float)A.py, vex = (float)A.vx, vey = (float)A.vy;
    const float re = (float)((1 + 15e-2) * A.r);
    double gx = A.gx - A.px, gy = A.gy - A.py;
    double sc = A.pref / norm2(gx, gy);
    const float pvx = (float)(sc * gx), pvy = (float)(sc * gy);
    const float max_speed = (float)A.pref;
    const float time_step = (float)dt, inv_th = 1.0f / 5.0f;
    const float c = (float)A.coop;

    // neighbour selection: nearest first, ties in index order, at most 10 (Agent::insertAgentNeighbor).
    // rank by counting == the insertion sort's result; ranks >= 10 are dropped.
    int nn = (n - 1) < CAGYM_MAXNB ? (n - 1) : CAGYM_MAXNB;
    for (int j = 0; j < n; j++) {
        if (j == i) continue;
        float ox = (float)T.px[base + j], oy = (float)T.py[base + j];
        float dx = pex - ox, dy = pey - oy;
        float dsq = dx * dx + dy * dy;
        int rank = 0;
        for (int l = 0; l < n; l++) {
            if (l == i || l == j) continue;
            float qx = pex - (float)T.px[base + l], qy = pey - (float)T.py[base + l];
            float qsq = qx * qx + qy * qy;
            rank += (qsq < dsq) || (qsq == dsq && l < j);
        }
        if (rank >= CAGYM_MAXNB) continue;
        float rpx = ox - pex, rpy = oy - pey;
        float rvx = vex - (float)T.vx[base + j], rvy = vey - (float)T.vy[base + j];
        float ro = (float)((1 + 15e-2) * T.r[base + j]);
        float d2 = rpx * rpx + rpy * rpy;
        float cr = re + ro, crsq = cr * cr;
        float ux, uy;
        float4 ln;
        if (d2 > crsq) {
            float wx = rvx - inv_th * rpx, wy = rvy - inv_th * rpy;
            float wlsq = wx * wx + wy * wy;
            float dp1 = wx * rpx + wy * rpy;
            if (dp1 < 0.0f && dp1 * dp1 > crsq * wlsq) {
                float wl = sqrtf(wlsq);
                float inv = 1.0f / wl;
                float uwx = wx * inv, uwy = wy * inv;
                ln.z = uwy;
                ln.w = -uwx;
                float s = cr * inv_th - wl;
                ux = s * uwx;
                uy = s * uwy;
            } else {
                float leg = sqrtf(d2 - crsq);
                float inv = 1.0f / d2;
                if (detf(rpx, rpy, wx, wy) > 0.0f) {
                    ln.z = (rpx * leg - rpy * cr) * inv;
                    ln.w = (rpx * cr + rpy * leg) * inv;
                } else {
                    ln.z = -((rpx * leg + rpy * cr) * inv);
                    ln.w = -((-rpx * cr + rpy * leg) * inv);
                }
                float dp2 = rvx * ln.z + rvy * ln.w;
                ux = dp2 * ln.z - rvx;
                uy = dp2 * ln.w - rvy;
            }
        } else {
            float inv_ts = 1.0f / time_step;
            float wx = rvx - inv_ts * rpx, wy = rvy - inv_ts * rpy;
            float wl = sqrtf(wx * wx + wy * wy);
            float inv = 1.0f / wl;
            float uwx = wx * inv, uwy = wy * inv;
            ln.z = uwy;
            ln.w = -uwx;
            float s = cr * inv_ts - wl;
            ux = s * uwx;
            uy = s * uwy;
        }
        ln.x = vex + c * ux;
        ln.y = vey + c * uy;
        L[rank * CAGYM_WAVE + lane] = ln;
    }
    float nvx, nvy;
    int fail = orca_lp2(L, lane, nn, max_speed, pvx, pvy, false, nvx, nvy);
    if (fail < nn) orca_lp3(L, P, lane, nn, fail, max_speed, nvx, nvy);
    float npx = pex + nvx * time_step, npy = pey + nvy * time_step;  // Agent::update, fp32
    double dpx = (double)npx - A.px, dpy = (double)npy - A.py;       // RVOPolicy.py:91-106, fp64
    double ang1 = atan2(dpy, dpx);
    double nh = fmod(ang1, 2 * kPi);
    if (nh < 0) nh += 2 * kPi;
    double dh = wrap_angle(nh - A.h);
    double speed = 1 / dt * norm2(dpx, dpy);
    if (fabs(dh) > kPi / 6) {
        dh = (dh > 0 ? 1.0 : (dh < 0 ? -1.0 : 0.0)) * (kPi / 6);
        speed = 0.;
    }
    out_speed = speed;
    out_dh = dh;
}
