// orca_lp_upfront.h -- MEASURED ALTERNATIVE, NOT PART OF THE PRODUCT (moved out of csrc/cagym_orca.h and csrc/cagym_kernels3.h in
// round 4; last tree that compiled it with -DCAGYM_LP_UPFRONT: commit c68e14a).  Two pieces that were measured together:
//   * orca_lp_upfront<GW, NL>: linearProgram2/3 with ALL linearProgram1 results computed up front and a find-first-set walk;
//   * lp_rank_lines3<MT, GW>: "lazy ranking" - half_planes3 stored the half-planes unsorted (row = neighbour slot with the ego's own
//     slot skipped: W.sorted[(q.j < q.i ? q.j : q.j - 1) * AS + a]) and only a busy ego's LP group ranked them.
// Bit-identical to the scan-and-jump solver (lp_upfront_check.c in this directory: 3 M half-plane sets on the CPU; the GPU parity
// tests of round 3), and SLOWER: 4096 x 10, 512-step launches 4.08 -> 5.44 ms; evidence in profiles/r3/valu_breakdown_lp_upfront.txt,
// profiles/r3/wave_trace_lp_upfront.txt.  To A/B it again: include this file behind cagym_orca.h, give cagym_lpl3 3 (4 when
// cagym_two3) float4 per lane for the free-space kernels, switch cagym_dsq_aliased off and call the two functions from phase C.
#pragma once

// ---- linearProgram2 + linearProgram3 of one ego on a GW-lane group: ALL linearProgram1 results up front ---------------------
// MEASURED ALTERNATIVE, NOT THE DEFAULT (-DCAGYM_LP_UPFRONT selects it together with the lazy ranking of cagym_kernels3.h;
// bit-identical to the scan-and-jump solver above: same GPU tests).  Round 3 built it as the review asked and measured it:
// 4096 x 10, 512-step launches 4.08 -> 5.44 ms, wave-VALU instructions per workgroup-step 3418 -> 4076 (ORCA part 1733 ->
// 2383, profiles/r3/valu_breakdown_*.txt), first program of an LP wave 3100 -> 8000 cycles (profiles/r3/wave_trace_lp_upfront.txt).
// Why: under four co-resident workgroups a wave issues one VALU instruction per ~12 cycles whatever its ILP, so a chain costs
// its INSTRUCTION COUNT, and computing R_i for all 9 lines (36 clips with a correctly rounded division each, ~500 instructions
// per pass, one more pass per linearProgram3 outer iteration for the whole wave) is more instructions than the 2.7 rounds
// x 70 the median ego needs.  Kept for the record and for A/B.
// (CPU proof of equivalence with the sequential programs, bit for bit on 3 million half-plane sets:
// tools/lp_upfront_check.c.)  linearProgram1(i) - "the optimum on line i subject to the lines before it" - does not
// depend on the running result, only on line i, the lines before it and the optimisation velocity.  The scan-and-jump
// solver of round 2 evaluated it when the walk reached line i: one lockstep round of ~1000 cycles of DEPENDENT work
// per projection (fetch the line, sqrt, one division per lane, two DPP reductions, a ballot), 2.7 rounds for the median
// ego and three times that in a crowd - the chain that ended every launch.  Here R_i and "infeasible at i" are computed
// for ALL lines first: lane j owns lines hi = n-1-j and lo = j (n-1 clips per lane whatever j is: the triangle of (line, earlier
// line) pairs is dealt evenly), each lane folds its own min / max in registers - independent, pipelined divisions, no
// cross-lane operation at all - and publishes (R_i, feasible_i) in the group's scratch.  linearProgram2's sequential part
// is then: every lane tests its two lines against the current point, a ballot turns that into a bit mask over line indices,
// find-first-set picks the next violated line, one LDS read fetches its R_i.  linearProgram3 keeps its outer scan; each
// inner program (directionOpt, projected lines with holes) runs through the SAME solver body: the function is a small
// state machine whose loop contains the solver once.
// Line set: L[k * stride] for k < n (the ego's column of the sorted half-planes).  S: group-private scratch of 2 NL - 1
// float4 (NL = compile-time bound on n): S[0 .. NL-1] = (R.x, R.y, feasible, -) per line, S[NL ..] = projected lines.
// (sx, sy) = linearProgram2's start (the optimisation velocity clipped to the disc: W.lpc).  All lanes of a group pass the same
// (n, radius, ox, oy, sx, sy); groups of one wave may differ.
template <int GW>
__device__ __forceinline__ uint32_t orca_line_mask(bool v_lo, bool v_hi, int nl, int gbase) {
    // lane j's `lo` bit belongs to line j, its `hi` bit to line nl-1-j: mask over line indices of the group
    const uint32_t gbits = (GW >= 32) ? 0xffffffffu : ((1u << GW) - 1u);
    const uint32_t blo = (uint32_t)(__ballot(v_lo) >> gbase) & gbits;
    const uint32_t bhi = (uint32_t)(__ballot(v_hi) >> gbase) & gbits;
    return blo | (__brev(bhi) >> ((32 - nl) & 31));
}
// one clip of linearProgram1: line li against the earlier line lk, folded into the lane's own (tLeft, tRight, infeasible)
__device__ __forceinline__ void orca_clip_acc(const float4 li, const float4 lk, bool take, float& tl, float& tr, bool& bad) {
    const float den = detf(li.z, li.w, lk.z, lk.w);
    const float num = detf(lk.z, lk.w, li.x - lk.x, li.y - lk.y);
    const bool par = fabsf(den) <= RVO_EPS;
    const float t = num / den;
    if (take) {
        if (par) bad = bad || (num < 0.0f);
        else if (den >= 0.0f) tr = fminf(tr, t);
        else tl = fmaxf(tl, t);
    }
}
// the rest of linearProgram1 once the interval is known
__device__ __forceinline__ float2 orca_lp1_point(const float4 l, float tl, float tr, float ox, float oy, bool dir_opt) {
    float t;
    if (dir_opt) {
        t = (ox * l.z + oy * l.w > 0.0f) ? tr : tl;
    } else {
        t = l.z * (ox - l.x) + l.w * (oy - l.y);
        if (t < tl) t = tl;
        else if (t > tr) t = tr;
    }
    return make_float2(l.x + t * l.z, l.y + t * l.w);
}
#ifndef CAGYM_LP_UNROLL1
#define CAGYM_LP_UNROLL1
#define CAGYM_LP_UNROLL2
#endif
template <int GW, int NL>
__device__ inline void orca_lp_upfront(const float4* L, int stride, float4* S, int j, int n, float radius, float ox, float oy,
                                       float sx, float sy, float& rx, float& ry, int* lp3_flag = nullptr, unsigned long long* wt = nullptr) {
#ifdef CAGYM_WAVETRACE
#define LPWT(k) do { if (wt && (threadIdx.x & 63) == 0) wt[(k) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LPWT(k) do { } while (0)
#endif
    const int gbase = (threadIdx.x & 63) & ~(GW - 1);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* PJ = S + NL;
    // the program being solved: line k = base[k * st], k < nl, existing when bit k of hv
    const float4* base = L;
    int st = stride, nl = n;
    uint32_t hv = 0xffffffffu;
    float optx = ox, opty = oy, stx = sx, sty = sy;
    bool dir_opt = false;
    // linearProgram3's outer state
    bool active = true, in_lp3 = false;
    float distance = 0.0f;
    int cur = 0;
    float4 li = zero4;
    rx = sx;
    ry = sy;
    LPWT(13);
    for (int pass = 0;; pass++) {
        if (pass == 1) LPWT(14);
        if (__ballot(active) == 0ull) break;  // wave-uniform: every group of the wave is done
        int fail = nl;
        float X = stx, Y = sty;
        if (active) {
            // ---- own lines and their chords in the disc ---------------------------------------------------------------
            const int hi = nl - 1 - j, lo = j;
            const bool hi_ok = j < ((nl + 1) >> 1) && ((hv >> hi) & 1u), lo_ok = j < (nl >> 1) && ((hv >> lo) & 1u);
            const float4 lh = base[(hi_ok ? hi : 0) * st], ll = base[(lo_ok ? lo : 0) * st];  // (a lane without a line reads row 0: never used)
            const float doth = lh.x * lh.z + lh.y * lh.w, dotl = ll.x * ll.z + ll.y * ll.w;
            const float disch = doth * doth + radius * radius - (lh.x * lh.x + lh.y * lh.y);
            const float discl = dotl * dotl + radius * radius - (ll.x * ll.x + ll.y * ll.y);
            const float sqh = sqrtf(disch < 0.0f ? 0.0f : disch), sql = sqrtf(discl < 0.0f ? 0.0f : discl);
            float tlh = -doth - sqh, trh = -doth + sqh, tll = -dotl - sql, trl = -dotl + sql;
            bool badh = !hi_ok || disch < 0.0f, badl = !lo_ok || discl < 0.0f;
            // ---- all clips: line hi by the lines 0 .. hi-1, line lo by 0 .. lo-1 -------------------------------------------
            CAGYM_LP_UNROLL1
            for (int k = 0; k < NL - 1; k++) {
                const bool take = hi_ok && k < hi;
                if (__ballot(take) == 0ull) break;
                const float4 lk = base[k * st];
                orca_clip_acc(lh, lk, take && ((hv >> k) & 1u), tlh, trh, badh);
            }
            CAGYM_LP_UNROLL2
            for (int k = 0; k < NL / 2 - 1; k++) {
                const bool take = lo_ok && k < lo;
                if (__ballot(take) == 0ull) break;
                const float4 lk = base[k * st];
                orca_clip_acc(ll, lk, take && ((hv >> k) & 1u), tll, trl, badl);
            }
            const bool feash = !badh && !(tlh > trh), feasl = !badl && !(tll > trl);
            const float2 Rh = orca_lp1_point(lh, tlh, trh, optx, opty, dir_opt), Rl = orca_lp1_point(ll, tll, trl, optx, opty, dir_opt);
            if (hi_ok) S[hi] = make_float4(Rh.x, Rh.y, feash ? 1.0f : 0.0f, 0.0f);
            if (lo_ok) S[lo] = make_float4(Rl.x, Rl.y, feasl ? 1.0f : 0.0f, 0.0f);
            // ---- the walk of linearProgram2: next violated line after c, by find-first-set ------------------------------------
            int c = -1;
            bool done = false;
            for (;;) {
                const bool vh = !done && hi_ok && hi > c && detf(lh.z, lh.w, lh.x - X, lh.y - Y) > 0.0f;
                const bool vl = !done && lo_ok && lo > c && detf(ll.z, ll.w, ll.x - X, ll.y - Y) > 0.0f;
                const uint32_t m = orca_line_mask<GW>(vl, vh, nl, gbase);
                if (!done) {
                    if (m == 0u) {
                        done = true;
                    } else {
                        const int i = __ffs((int)m) - 1;
                        const float4 e = S[i];  // same wave: the stores above are ordered before this load
                        if (e.z == 0.0f) { fail = i; done = true; }  // infeasible at line i: the result keeps its value (tempResult)
                        else { c = i; X = e.x; Y = e.y; }
                    }
                }
                if (__ballot(!done) == 0ull) break;
            }
        }
        // ---- linearProgram3 around it ------------------------------------------------------------------------------------------
        if (active) {
            if (!in_lp3) {
                rx = X;
                ry = Y;
                if (fail == n) {
                    active = false;
                } else {
                    in_lp3 = true;
                    distance = 0.0f;
                    cur = fail;
                    if (lp3_flag && j == 0) *lp3_flag = 1;  // this workgroup is in a crowd
                }
            } else {
                if (fail == nl) { rx = X; ry = Y; }  // the inner program was feasible; otherwise the result keeps its value
                distance = detf(li.z, li.w, li.x - rx, li.y - ry);
            }
        }
        if (active) {  // in linearProgram3: the next ORIGINAL line violated by more than `distance`
            const int hi0 = n - 1 - j, lo0 = j;
            const bool h_ok = j < ((n + 1) >> 1), l_ok = j < (n >> 1);
            const float4 oh = L[(h_ok ? hi0 : 0) * stride], ol = L[(l_ok ? lo0 : 0) * stride];
            const bool wh = h_ok && hi0 >= cur && detf(oh.z, oh.w, oh.x - rx, oh.y - ry) > distance;
            const bool wl = l_ok && lo0 >= cur && detf(ol.z, ol.w, ol.x - rx, ol.y - ry) > distance;
            const uint32_t wm = orca_line_mask<GW>(wl, wh, n, gbase);
            if (wm == 0u) {
                active = false;
            } else {
                const int i = __ffs((int)wm) - 1;
                cur = i + 1;
                li = L[i * stride];
                // the lines before i projected onto line i (a hole where "parallel, same direction")
                bool hh = false, hl = false;
                float4 pj;
                if (h_ok && hi0 < i) {
                    hh = orca_project(li, oh, pj);
                    if (hh) PJ[hi0] = pj;
                }
                if (l_ok && lo0 < i) {
                    hl = orca_project(li, ol, pj);
                    if (hl) PJ[lo0] = pj;
                }
                hv = orca_line_mask<GW>(hl, hh, n, gbase);
                nl = i;
                base = PJ;
                st = 1;
                optx = -li.w;
                opty = li.z;
                dir_opt = true;
                stx = optx * radius;
                sty = opty * radius;
            }
        }
    }
}

// Lazy ranking (free-space kernels): the LP group of busy ego `a` (slot sl of a world of nw agents) turns its column of
// unsorted half-planes (row q <-> neighbour slot q < sl ? q : q + 1) into nearest-first order IN PLACE: lane j takes the
// candidates q = j and q = j + GW, counts each one's rank in the ego's row of squared distances (Agent::insertAgentNeighbor's
// order: nearer first, ties by lower index) and stores it to row `rank` when that is below nn = min(nw - 1, maxNeighbors).
// One wave: its LDS operations complete in program order, so every candidate is read before any row is overwritten.
template <int MT, int GW>
__device__ __forceinline__ void lp_rank_lines3(const Lds3& W, int a, int sl, int nw, int nn, int j, int M, int MP, int AS) {
    const int q0 = j, q1 = j + GW;
    const int o0 = q0 < sl ? q0 : q0 + 1, o1 = q1 < sl ? q1 : q1 + 1;
    const bool e0 = q0 < M - 1 && o0 < nw, e1 = q1 < M - 1 && o1 < nw;
    // (clamped addresses instead of conditional loads: a lane without a candidate reads row 0 and never uses it)
    const float4 c0 = W.sorted[(e0 ? q0 : 0) * AS + a], c1 = W.sorted[(e1 ? q1 : 0) * AS + a];
    const float d0 = __uint_as_float(W.dsq[a * MP + (e0 ? o0 : 0)].y), d1 = __uint_as_float(W.dsq[a * MP + (e1 ? o1 : 0)].y);
    const int r0 = neighbour_rank3<MT>(W, a, o0, d0, MP);
    int r1 = 0;
    if (__ballot(e1) != 0ull) r1 = neighbour_rank3<MT>(W, a, o1, d1, MP);
    asm volatile("" ::: "memory");  // all reads of the column stay ahead of its rewriting
    if (e0 && r0 < nn) W.sorted[r0 * AS + a] = c0;
    if (e1 && r1 < nn) W.sorted[r1 * AS + a] = c1;
}

