// cagym_kernels3.h -- software-pipelined env.step() kernels (generation 3, the default).
//
// Why (profiles/r2/wgtrace_*.txt, stamps_gen2_*.txt): a 4096 x 10 batch is 16 worlds per CU.  A generation-2 workgroup
// alone on its CU still needs 8.3 us per step and four co-resident ones 10.1 us: the launch is bound by the LENGTH OF
// THE DEPENDENT CHAIN of one step (six phases separated by barriers, the per-agent phases on one wave with the other
// three idle), not by throughput.  Only S1(t) -> half-planes(t+1) -> linear programs(t+1) -> S1(t+1) is a true cycle;
// everything else of a step is moved off it:
//
//   phase C   linear programs of step t on the first waves (8-lane groups)    ||  OAS rows of step t-1 on the idle waves,
//   phase D   S1(t) on wave 0: action maps + dynamics, result kept in VGPRs   ||  64-row chunks claimed from an LDS counter
//             barrier; wave 0 publishes the moved agents to LDS; barrier
//   phase A   pair distances / collisions / OAS keys of step t + the fp32 squared distances the next ORCA ranking needs
//             (one lane per unordered pair); last wave: ego frames + preferred velocities
//   phase B   S2(t) on wave 0: rewards, done, game_over, auto-reset            ||  half-planes of step t+1 on waves 1.., each
//             pair lane also ranks its two half-planes into nearest-first order (the LP groups start from sorted lines)
//   (rare)    a world was reset in S2: keys, distances, preferred velocities and half-planes are rebuilt in two extra phases
//
// Wave 0 waits for the other LP waves on an LDS counter (release/acquire at workgroup scope), not on a barrier, so the
// row workers never stop between C and D.  Arithmetic is shared with generation 1 (cagym_device.h, cagym_orca.h):
// both produce bit-identical results (tests/test_hip_parity.py); generation 2 (phase-split, one barrier-separated phase after
// the other) was retired once generation 3 reproduced it bit for bit.
#pragma once
#include "cagym_kernels.h"

#include "cagym_spin.h"   // lds_wait_ge: the bounded intra-workgroup wait
#include "cagym_trace.h"  // STAMP / WGTRACE / WAVETRACE / PMARK: diagnostic hooks, empty in the shipped library

#ifndef CAGYM_GW10
#define CAGYM_GW10 8  // lanes per ORCA LP group when M <= 10 (nn <= 9 half-planes)
#endif

__host__ __device__ inline size_t a16(size_t x) { return (x + 15) & ~(size_t)15; }

// Issue priorities.  A workgroup whose worlds needed linearProgram3 in the previous step is in a crowd and stays there for many
// steps: its step is the long one and the launch ends with the slowest workgroup, so its waves take issue priority over the
// co-resident workgroups (which have slack).  Inside a workgroup the observation rows are off the critical chain: their waves
// run at priority 0, the chain (linear programs, S1, pair phases) at 1.  (-DCAGYM_NO_LAG_PRIORITY: A/B switch, 20-step launch 264 -> 233 us.)
__device__ __forceinline__ void prio_chain3(bool lagging) {
#ifndef CAGYM_NO_LAG_PRIORITY
    if (lagging) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(1);
#endif
}
__device__ __forceinline__ void prio_rows3() {
#ifndef CAGYM_NO_LAG_PRIORITY
    __builtin_amdgcn_s_setprio(0);
#endif
}
// LP group geometry of a specialisation (compile-time M = MT, 0 = run-time M): lanes per group, "more than GW + 1 half-planes
// possible", and the group's scratch in units of GW float4 (2 GW projected lines; the OBST instantiation's orca_lp_group_n: 4 GW)
__host__ __device__ constexpr int cagym_gw3(int MT) { return MT > 0 && MT <= 5 ? 4 : (MT > 0 && MT <= 10 ? CAGYM_GW10 : 16); }
__host__ __device__ constexpr bool cagym_two3(int MT) { return !(MT > 0 && MT - 1 <= cagym_gw3(MT) + 1); }
__host__ __device__ constexpr int cagym_lpl3(int MT, bool obst) { return obst ? 4 : 2; }
__host__ __device__ inline int cagym_mp(int M) { return (M + 3) & ~3; }

// AS = agent slots per workgroup (worlds per workgroup x M, rounded up to 4); 64 when a full wave is used
__host__ __device__ inline int cagym_as(int M, int wpw) { return wpw > 0 ? ((wpw * M + 3) & ~3) : 64; }


__device__ __forceinline__ LaneCtx make_ctx2(const CagymDev& D, int M, int wpw) {
    LaneCtx C;
    C.lane = threadIdx.x;
    C.wpw = wpw;
    C.wl = C.lane / M;
    C.slot = C.lane - C.wl * M;
    C.base = C.wl * M;
    C.world = blockIdx.x * C.wpw + C.wl;
    int rem = D.N - (int)blockIdx.x * C.wpw;
    C.worlds_valid = rem < C.wpw ? rem : C.wpw;
    C.valid = C.wl < C.wpw && C.world < D.N;
    C.n = C.valid ? D.n_agents[C.world] : 0;
    C.episode = C.valid ? D.episode[C.world] : 0;
    C.active = C.valid && C.slot < C.n;
    return C;
}

// pair slot p = agent * M + j  ->  (agent lane a, neighbour slot j, world_local wl, agent slot sl)
struct PairIdx {
    int a, j, wl, sl;
};
__device__ __forceinline__ PairIdx pair_of(int p, int M, uint32_t inv_m) {
    PairIdx q;
    q.a = (int)__umulhi((uint32_t)p, inv_m);  // p / M for p < 2^16 (inv_m = 2^32 / M + 1)
    q.j = p - q.a * M;
    q.wl = (int)__umulhi((uint32_t)q.a, inv_m);
    q.sl = q.a - q.wl * M;
    return q;
}

// Unordered pairs of M slots by circular difference: (i, i+k mod M) for k = 1..(M-1)/2, plus the M/2 diameters
// when M is even.  p -> (world of the workgroup, i, j) with compile-time divisors only.
struct UPair {
    int wl, i, j;
};
template <int MT>
struct UnorderedPairs {
    static constexpr int N = MT > 0 ? MT * (MT - 1) / 2 : 1;
    static constexpr int H = MT > 0 ? (MT - 1) / 2 : 1;
    static constexpr int MM = MT > 0 ? MT : 1;
    __device__ static __forceinline__ UPair of(int p) {
        UPair q;
        q.wl = p / N;
        const int u = p - q.wl * N;
        if (u < MM * H) {
            const int k = u / MM;
            q.i = u - k * MM;
            q.j = q.i + k + 1;
            if (q.j >= MM) q.j -= MM;
        } else {
            q.i = u - MM * H;
            q.j = q.i + MM / 2;
        }
        return q;
    }
};


struct Lds3 {
    double *tpx, *tpy, *tvx, *tvy, *tr, *tprx, *tpry;
    double *th, *the, *tdg, *ttrem, *tt, *tgx, *tgy, *tpref, *tspeed, *tdh, *taux0, *taux1, *tcoopd;
    float2* tact;
    float* tcoop;
    uint32_t* tst;
    int* tstep;
    int* tmoved;   // [AS] the agent moved in this step's S1 (ego frame still to be updated)
    int* trvo;     // [AS] live RVO ego as of the last S1 (the next step's half-planes are built beside S2)
    int* nobl;     // [AS] obstacle half-planes of the ego (rows 0 .. nobl-1 of its column of `sorted`)
    int* wn;       // [32] agents per world of this workgroup
    int* wsc;      // [32] scenario of the world's current episode (rectangles, raster)
    int* wnob;     // [32] rectangles of that scenario
    float4* rect;  // [worlds x Kobs x 4] the worlds' prepared rectangles (OBST only), staged at episode start
    uint16_t* blist;  // [AS x 16] u16 + [AS x 16] u8 + [16] u8 (OBST only) the workgroup's list of the laser beams that can meet a rectangle, their sample ranges, beams per pass
    uint32_t* cov;    // [ko][AS] (OBST only) obstacle_lines_phase3's coverage bit matrix
    int* wall;        // [4 AS] (OBST only) wall_prep3's per-agent cell / window / row half-widths
    int* flag;     // [16] 0: a world was reset this step   1: OAS row chunks claimed   2: LP waves finished
                   //      3/4: some ego needed linearProgram3 this / the previous step   5/6: obstacle_lines_phase3's work list
                   //      7, 8, 10: LaserScan: slab-test passes claimed / finished, sampling rounds claimed
    float2* lpv;   // [AS] preferred (optimisation) velocity of each ego
    float2* lpc;   // [AS] pref velocity clipped to maxSpeed = LP start; LP result afterwards
    float* lpr;    // [AS] maxSpeed of the ego (LP radius)
    int* lpk;      // [AS] compact list of the busy egos
    int* busy;     // [AS] some half-plane of the ego is violated by its LP start
    uint2* dsq;      // [AS*MP]       {slot, bits of the fp32 squared centre distance ego->slot as RVO2 computes it (+inf = no such
                     //               neighbour)}: ONE 64-bit key per neighbour whose unsigned order is Agent::insertAgentNeighbor's
                     //               (nearer first, ties by lower index; distances are >= 0, so float order = bit order)
    float4* sorted;  // [ko + M - 1][AS]  half-planes in solve order: rows < ko obstacle lines (ko = 2 rectangles' worth per
                     //                  rectangle, 0 without obstacles), rows ko + rank agent lines nearest-first
    float4* lp3;     // [lpl NT]      linearProgram3 scratch, lpl GW entries per LP group (lpl = 2, 4 with obstacles); the
                     //               obstacle-neighbour sort of phase A borrows it
    double* keys;    // [AS*MP]       OAS sort key (-inf = not observed)
    unsigned long long* gmin;  // [AS]    min over the agent's pairs (as the LOWER index, Q7) of d - (r_i + r_j), as an order-preserving
                     //               64-bit key: every pair lane folds its gap in with ONE LDS ds_min_u64 (a gap MATRIX [AS*MP] and a row
                     //               scan in S2 before round 3: 3.8 KB of LDS that kept the 5-worlds-per-workgroup variant at 3 per CU)
    uint8_t* hit;    // [AS*MP]       pair collides
};

// The ORCA neighbour keys are only alive between phase A (pair lanes write them) and phase B (the half-plane lanes rank with them); the
// LP scratch only in phase C.  The free-space kernels therefore keep both in the same LDS bytes (M = 20: 43.5 -> 37.1 KB, a fourth
// workgroup per CU); the OBST kernels borrow the scratch for their obstacle-neighbour lists in phase A and keep them apart.
// Only where it buys a workgroup (two half-planes per LP lane: M = 20 and the run-time-M kernels): the M <= 10 kernels sit exactly at 128
// VGPRs and the three extra stores per step cost them a spill.
__host__ __device__ constexpr bool cagym_dsq_aliased(bool obst, int MT) { return !obst && cagym_two3(MT); }
__host__ __device__ inline size_t cagym_lds3_head(int AS) {
    return (size_t)20 * AS * 8 + AS * 8 + (size_t)6 * AS * 4 + 96 * 4 + 16 * 4 + (size_t)2 * AS * 8 + (size_t)3 * AS * 4;
}
// The OBST instantiation's phase-A-only arrays live in bytes that are dead in phase A (cfg4: 61.1 -> 53.4 KB, a THIRD workgroup per CU:
// env kernel 229 -> 17x us, profiles/r3/cfg4_occupancy_ab.txt).  Dead between the linear programs (phase C) and the half-plane lanes
// (phase B): the agent rows [ko, ko + M - 1) of `sorted` - they take the coverage bits and the wall-test records - and, directly behind
// them, the LP scratch - it already lends its head to the obstacle-neighbour lists (13 B per (ego, candidate)) and now its tail to the
// neighbour keys `dsq` (alive from phase A's pair lanes to phase B's ranking; the scratch is phase C's).  Only when everything fits.
__host__ __device__ inline bool cagym_obst_alias(int M, int AS, int NT, int ko, int lpl) {
    const size_t MP = cagym_mp(M);
    return a16((size_t)ko * AS * 4) + (size_t)AS * 16 <= (size_t)(M - 1) * AS * 16 &&
           a16((size_t)13 * ko * AS) + (size_t)AS * MP * 8 <= (size_t)lpl * NT * 16;
}
// ko: rows of `sorted` reserved for obstacle lines; lpl: LP scratch per group in units of GW float4 (cagym_lpl3); obst: the OBST
// instantiation's extra arrays
__host__ __device__ inline size_t cagym_lds3_bytes(int M, int AS, int NT, int ko, int lpl, bool obst, int MT) {
    const size_t MP = cagym_mp(M);
    const bool oa = obst && cagym_obst_alias(M, AS, NT, ko, lpl);
    // (free space: the neighbour keys `dsq` live in the LP scratch - written in phase A, read in phase B, the scratch is phase C's)
    return a16(cagym_lds3_head(AS)) + ((cagym_dsq_aliased(obst, MT) || oa) ? 0 : a16(AS * MP * 8)) + (size_t)(ko + M - 1) * AS * 16 + (size_t)lpl * NT * 16 + AS * MP * 8 + (size_t)AS * 8 + a16(AS * MP) +
           (size_t)(AS / M) * (ko / 2) * 64 +  // staged rectangles: worlds x Kobs x 4 float4
           (obst ? a16((size_t)AS * 48 + 16) + (oa ? 0 : a16((size_t)ko * AS * 4) + (size_t)AS * 16) : 0);  // OBST: beam list (+ coverage bits, wall prep)
}

__device__ __forceinline__ Lds3 carve_lds3(unsigned char* smem, int M, int AS, int NT, int ko, int lpl, bool obst, int MT) {
    Lds3 W;
    const size_t MP = cagym_mp(M);
    W.tpx = reinterpret_cast<double*>(smem);
    W.tpy = W.tpx + AS; W.tvx = W.tpy + AS; W.tvy = W.tvx + AS; W.tr = W.tvy + AS; W.tprx = W.tr + AS; W.tpry = W.tprx + AS;
    W.th = W.tpry + AS; W.the = W.th + AS; W.tdg = W.the + AS; W.ttrem = W.tdg + AS; W.tt = W.ttrem + AS;
    W.tgx = W.tt + AS; W.tgy = W.tgx + AS; W.tpref = W.tgy + AS; W.tspeed = W.tpref + AS; W.tdh = W.tspeed + AS;
    W.taux0 = W.tdh + AS; W.taux1 = W.taux0 + AS; W.tcoopd = W.taux1 + AS;
    W.tact = reinterpret_cast<float2*>(W.tcoopd + AS);
    W.lpv = W.tact + AS;
    W.lpc = W.lpv + AS;
    W.tcoop = reinterpret_cast<float*>(W.lpc + AS);
    W.tst = reinterpret_cast<uint32_t*>(W.tcoop + AS);
    W.tstep = reinterpret_cast<int*>(W.tst + AS);
    W.tmoved = W.tstep + AS;
    W.trvo = W.tmoved + AS;
    W.lpr = reinterpret_cast<float*>(W.trvo + AS);
    W.lpk = reinterpret_cast<int*>(W.lpr + AS);
    W.busy = W.lpk + AS;
    W.nobl = W.busy + AS;
    W.wn = W.nobl + AS;
    W.wsc = W.wn + 32;
    W.wnob = W.wsc + 32;
    W.flag = W.wnob + 32;
    const bool oa = obst && cagym_obst_alias(M, AS, NT, ko, lpl);
    unsigned char* u = smem + a16(cagym_lds3_head(AS));
    if (!cagym_dsq_aliased(obst, MT) && !oa) {
        W.dsq = reinterpret_cast<uint2*>(u);
        u += a16(AS * MP * 8);
    }
    W.sorted = reinterpret_cast<float4*>(u);
    u += (size_t)(ko + M - 1) * AS * 16;
    W.lp3 = reinterpret_cast<float4*>(u);
    if (cagym_dsq_aliased(obst, MT)) W.dsq = reinterpret_cast<uint2*>(u);  // AS * MP * 8 <= lpl * NT * 16 for every specialisation (checked by cagym_create)
    if (oa) W.dsq = reinterpret_cast<uint2*>(u + a16((size_t)13 * ko * AS));  // behind obstacle_lines_phase3's lists
    u += (size_t)lpl * NT * 16;
    W.keys = reinterpret_cast<double*>(u);
    W.gmin = reinterpret_cast<unsigned long long*>(W.keys + AS * MP);
    W.hit = reinterpret_cast<uint8_t*>(W.gmin + AS);
    W.rect = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(W.hit) + a16(AS * MP));
    W.blist = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(W.rect) + (size_t)(AS / M) * (ko / 2) * 64);
    W.cov = oa ? reinterpret_cast<uint32_t*>(W.sorted + (size_t)ko * AS) : reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(W.blist) + a16((size_t)AS * 48 + 16));
    W.wall = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(W.cov) + a16((size_t)ko * AS * 4));
    return W;
}

// ---- the split step (cagym_step_begin / cagym_step_finish, cagym_split3.h) ------------------------------------------------------
// POST half (S1 with the actions of every policy in hand -> pair phase -> S2 + LaserScan -> observations): run_steps3 itself in
// its one-step form with no RVO work left to do, on a carve WITHOUT the half-plane rows, the LP scratch and the neighbour keys
// (cfg4: 53.4 -> 17 KB); krect = rectangles staged per world (the LaserScan's slab test), 0 = none
__host__ __device__ inline size_t cagym_lds3_post_bytes(int M, int AS, int krect, bool obst) {
    const size_t MP = cagym_mp(M);
    return a16(cagym_lds3_head(AS)) + AS * MP * 8 + (size_t)AS * 8 + a16(AS * MP) + (size_t)(AS / M) * krect * 64 +
           (obst ? a16((size_t)AS * 48 + 16) + (size_t)AS * 16 : 0);
}
__device__ __forceinline__ Lds3 carve_lds3_post(unsigned char* smem, int M, int AS, int krect) {
    Lds3 W = carve_lds3(smem, M, AS, 0, 0, 0, false, 1);  // the head's pointers (every size argument below the head is unused here)
    const size_t MP = cagym_mp(M);
    W.dsq = nullptr; W.sorted = nullptr; W.lp3 = nullptr; W.cov = nullptr;
    W.keys = reinterpret_cast<double*>(smem + a16(cagym_lds3_head(AS)));
    W.gmin = reinterpret_cast<unsigned long long*>(W.keys + AS * MP);
    W.hit = reinterpret_cast<uint8_t*>(W.gmin + AS);
    W.rect = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(W.hit) + a16(AS * MP));
    W.blist = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(W.rect) + (size_t)(AS / M) * krect * 64);
    W.wall = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(W.blist) + a16((size_t)AS * 48 + 16));
    return W;
}

// ---- the ONE-step launch among rectangles (k_step3, OBST): time-shared LDS ("OVL") ----------------------------------------------------
// A one-step launch runs its phases exactly once, in order: [rectangles, obstacle half-planes, neighbour keys, agent half-planes,
// linear programs] and then [S1, pair phase + wall test, S2 + LaserScan, observations].  What the first group keeps in LDS - the
// half-plane rows `sorted` and the scratch of the candidate lists / neighbour keys / projected lines - is dead once the linear
// programs are solved; what the second group needs - OAS keys, collision bits, gap minima, beam list, wall-test records - is not
// alive before.  They share bytes: cfg4 53.4 -> 39.8 KB = a FOURTH workgroup per CU, and its 2 048 workgroups are exactly two rounds
// instead of 2.67 (profiles/r4/cfg4_timeline.txt).  The scratch itself is one region for three consumers alive one after the other
// (cagym_split3.h's PRE half uses the same scheme): obstacle_lines_phase3's candidate lists (13 B per (ego, candidate)), then the
// neighbour keys `dsq`, then the LP groups' projected lines (2 GW float4 per group, orca_lp_group_n<.., PC>).
__host__ __device__ inline size_t cagym_ovl3_scratch(int M, int AS, int NT, int ko) {
    const size_t lists = a16((size_t)13 * ko * AS), keys = a16((size_t)AS * cagym_mp(M) * 8), proj = (size_t)2 * NT * 16;
    return lists > keys ? (lists > proj ? lists : proj) : (keys > proj ? keys : proj);
}
// the coverage bits of obstacle_lines_phase3 live in the agent rows of `sorted` (written by the half-plane lanes afterwards) when they fit
__host__ __device__ inline bool cagym_ovl3_cov_aliased(int M, int AS, int ko) { return a16((size_t)ko * AS * 4) <= (size_t)(M - 1) * AS * 16; }
__host__ __device__ inline size_t cagym_ovl3_late(int M, int AS) {  // keys, gmin, hit, beam list, wall records
    const size_t MP = cagym_mp(M);
    return (size_t)AS * MP * 8 + (size_t)AS * 8 + a16(AS * MP) + a16((size_t)AS * 48 + 16) + (size_t)AS * 16;
}
__host__ __device__ inline size_t cagym_lds3_ovl_bytes(int M, int AS, int NT, int ko) {
    const size_t early = (size_t)(ko + M - 1) * AS * 16 + cagym_ovl3_scratch(M, AS, NT, ko), late = cagym_ovl3_late(M, AS);
    return a16(cagym_lds3_head(AS)) + (size_t)(AS / M) * (ko / 2) * 64 + (early > late ? early : late) +
           (cagym_ovl3_cov_aliased(M, AS, ko) ? 0 : a16((size_t)ko * AS * 4));
}
__device__ __forceinline__ Lds3 carve_lds3_ovl(unsigned char* smem, int M, int AS, int NT, int ko) {
    Lds3 W = carve_lds3(smem, M, AS, 0, 0, 0, false, 1);  // the head's pointers
    const size_t MP = cagym_mp(M);
    unsigned char* u = smem + a16(cagym_lds3_head(AS));
    W.rect = reinterpret_cast<float4*>(u);
    u += (size_t)(AS / M) * (ko / 2) * 64;
    // early: rows, then the scratch
    W.sorted = reinterpret_cast<float4*>(u);
    W.lp3 = reinterpret_cast<float4*>(u + (size_t)(ko + M - 1) * AS * 16);
    W.dsq = reinterpret_cast<uint2*>(W.lp3);
    // late: the same bytes from the region's start
    W.keys = reinterpret_cast<double*>(u);
    W.gmin = reinterpret_cast<unsigned long long*>(W.keys + AS * MP);
    W.hit = reinterpret_cast<uint8_t*>(W.gmin + AS);
    W.blist = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(W.hit) + a16(AS * MP));
    W.wall = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(W.blist) + a16((size_t)AS * 48 + 16));
    const size_t early = (size_t)(ko + M - 1) * AS * 16 + cagym_ovl3_scratch(M, AS, NT, ko), late = cagym_ovl3_late(M, AS);
    W.cov = cagym_ovl3_cov_aliased(M, AS, ko) ? reinterpret_cast<uint32_t*>(W.sorted + (size_t)ko * AS)
                                              : reinterpret_cast<uint32_t*>(u + (early > late ? early : late));
    return W;
}

// the prepared rectangles (and their count) of every world of the workgroup: HBM -> LDS, all lanes; W.wsc must be visible
__device__ __forceinline__ void stage_rects3(const CagymDev& D, const Lds3& W, int wpw, int worlds_valid) {
    const int per = D.Kobs * 4;
    for (int q = threadIdx.x; q < wpw * per; q += blockDim.x) {
        const int wl = q / per;
        if (wl < worlds_valid) W.rect[q] = D.sc_obst_prep[(size_t)W.wsc[wl] * per + (q - wl * per)];
    }
    if ((int)threadIdx.x < wpw) W.wnob[threadIdx.x] = (int)threadIdx.x < worlds_valid ? D.sc_nobst[W.wsc[threadIdx.x]] : 0;
}

__device__ __forceinline__ void lds3_store_moved(const Lds3& W, const Agent& A, int lane) {
    W.tpx[lane] = A.px; W.tpy[lane] = A.py; W.tvx[lane] = A.vx; W.tvy[lane] = A.vy;
    W.tprx[lane] = A.prx; W.tpry[lane] = A.pry;
    W.th[lane] = A.h; W.the[lane] = A.he; W.tdg[lane] = A.dg; W.ttrem[lane] = A.trem; W.tt[lane] = A.t;
    W.tspeed[lane] = A.speed; W.tdh[lane] = A.dh; W.taux0[lane] = A.aux0; W.taux1[lane] = A.aux1;
    W.tact[lane] = make_float2(A.a0, A.a1);
    W.tst[lane] = A.st;
    W.tstep[lane] = A.step;
}
// what S1 (take_action<false>) changes: the ego frame (prx, pry, he, dg) is phase A's, the constants never change
__device__ __forceinline__ void lds3_store_s1(const Lds3& W, const Agent& A, int lane) {
    W.tpx[lane] = A.px; W.tpy[lane] = A.py; W.tvx[lane] = A.vx; W.tvy[lane] = A.vy;
    W.th[lane] = A.h; W.ttrem[lane] = A.trem; W.tt[lane] = A.t;
    W.tspeed[lane] = A.speed; W.tdh[lane] = A.dh; W.taux0[lane] = A.aux0; W.taux1[lane] = A.aux1;
    W.tact[lane] = make_float2(A.a0, A.a1);
    W.tst[lane] = A.st;
    W.tstep[lane] = A.step;
}
__device__ __forceinline__ void lds3_store_agent(const Lds3& W, const Agent& A, int lane) {
    lds3_store_moved(W, A, lane);
    W.tr[lane] = A.r; W.tgx[lane] = A.gx; W.tgy[lane] = A.gy; W.tpref[lane] = A.pref;
    W.tcoopd[lane] = A.coop;
    W.tcoop[lane] = (float)A.coop;
}
__device__ __forceinline__ Agent lds3_load_agent(const Lds3& W, int lane) {
    Agent A;
    A.px = W.tpx[lane]; A.py = W.tpy[lane]; A.vx = W.tvx[lane]; A.vy = W.tvy[lane]; A.r = W.tr[lane];
    A.prx = W.tprx[lane]; A.pry = W.tpry[lane];
    A.h = W.th[lane]; A.he = W.the[lane]; A.dg = W.tdg[lane]; A.trem = W.ttrem[lane]; A.t = W.tt[lane];
    A.gx = W.tgx[lane]; A.gy = W.tgy[lane]; A.pref = W.tpref[lane]; A.speed = W.tspeed[lane]; A.dh = W.tdh[lane];
    A.aux0 = W.taux0[lane]; A.aux1 = W.taux1[lane]; A.coop = W.tcoopd[lane];
    const float2 a = W.tact[lane];
    A.a0 = a.x; A.a1 = a.y;
    A.st = W.tst[lane];
    A.step = W.tstep[lane];
    return A;
}

// live RVO ego: solves an ORCA program in the next step.  S2 turns every at-goal / timed-out / collided agent into a
// done one, so testing the three flags after S1 already tells (only a collision found by the concurrent S2 is missed:
// that ego gets one wasted linear program, its result is never read).
__device__ __forceinline__ int live_rvo(uint32_t st, bool active) {
    return (active && ST_POLICY(st) == CAGYM_POL_RVO &&
            !(st & (CAGYM_FLAG_DONE | CAGYM_FLAG_AT_GOAL | CAGYM_FLAG_RAN_OUT_OF_TIME | CAGYM_FLAG_IN_COLLISION))) ? 1 : 0;
}

// prefVelocity / maxSpeed / LP start of agent a (publish_pref_velocity of generation 2 on the Lds3 carve)
__device__ __forceinline__ void publish_pref_velocity3(const Lds3& W, int a) {
    const double gx = W.tgx[a] - W.tpx[a], gy = W.tgy[a] - W.tpy[a];
    const double pref = W.tpref[a];
    const double sc = pref / norm2(gx, gy);
    const float ox = (float)(sc * gx), oy = (float)(sc * gy), radius = (float)pref;
    W.lpv[a] = make_float2(ox, oy);
    W.lpr[a] = radius;
    float cx = ox, cy = oy;
    if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        cx = ox * inv * radius;
        cy = oy * inv * radius;
    }
    W.lpc[a] = make_float2(cx, cy);
    W.busy[a] = 0;
}

// Wall test (env.py:656-666) of every agent slot of the workgroup in two parts.  wall_prep3: one lane per agent computes the
// cell, the window and the rows' half-widths into W.wall (4 ints per agent).  wall_rows3 (behind a barrier): a wave round
// takes 3 agents x 17 raster rows, a ballot folds the rows, the result waits in W.lpk (idle between two LP phases) for S2.
// On wave 0 alone, inside S2, the test was 15 000 of the step's 78 000 cycles with the other waves idle; with the per-agent
// part repeated on every row lane it was no faster.
__device__ __forceinline__ void wall_prep3(const CagymDev& D, const Lds3& W, int a, int ko, uint32_t inv_m, int M) {
    const int wl = (int)__umulhi((uint32_t)a, inv_m);
    int nrect;  // (an if / else, not `c ? lds : global`: the conditional operator would select between the two ADDRESSES - a flat load)
    if (ko > 0) nrect = W.wnob[wl];
    else nrect = D.sc_nobst[W.wsc[wl]];
    WallPrep P = {0, 0, 0ull};
    if ((a - wl * M) < W.wn[wl] && nrect > 0) P = wall_prep(W.tpx[a], W.tpy[a], W.tr[a]);
    reinterpret_cast<int4*>(W.wall)[a] = make_int4(P.cell, P.flags, (int)(P.w & 0xffffffffull), (int)(P.w >> 32));
}
template <int NWAVES>
__device__ __forceinline__ void wall_rows3(const CagymDev& D, const Lds3& W, int nagents, uint32_t inv_m) {
    constexpr int ROUNDS = (22 + NWAVES - 1) / NWAVES;  // <= 64 agent slots = 22 groups of 3
    const int lane = threadIdx.x & (CAGYM_WAVE - 1), wave = threadIdx.x / CAGYM_WAVE;
    const int sub = lane / 17, k = lane - sub * 17 - 8;  // lanes 51 .. 63 idle
    // every round's raster words are requested before the first ballot (straight-line: a group beyond the last agent reads row 0 of
    // the first world's raster and is ignored); round after round, each waited ~1 us for its two gathers
    bool hit_row[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int a = (wave + r * NWAVES) * 3 + sub;
        const bool slot_ok = sub < 3 && a < nagents;
        const int ac = slot_ok ? a : 0;
        const int4 P = reinterpret_cast<const int4*>(W.wall)[ac];
        const int wl = (int)__umulhi((uint32_t)ac, inv_m);
        const uint32_t* map = D.map_bits + (size_t)W.wsc[wl] * CAGYM_MAPD * CAGYM_MAPW;
        hit_row[r] = wall_row_hit(map, P.x, slot_ok ? P.y : 0, ((unsigned long long)(uint32_t)P.w << 32) | (uint32_t)P.z, k);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int a = (wave + r * NWAVES) * 3 + sub;
        const bool slot_ok = sub < 3 && a < nagents;
        const unsigned long long m = __ballot(hit_row[r]);
        if (slot_ok && k == -8) {
            bool hit = ((m >> (sub * 17)) & 0x1ffffull) != 0ull;
            const int4 P = reinterpret_cast<const int4*>(W.wall)[a];
            if (P.y & 2) {  // radius > 0.7 m: the whole test on this lane (rare)
                const int wl = (int)__umulhi((uint32_t)a, inv_m);
                hit = wall_collision(D.map_bits + (size_t)W.wsc[wl] * CAGYM_MAPD * CAGYM_MAPW, W.tpx[a], W.tpy[a], W.tr[a]);
            }
            W.lpk[a] = hit ? 1 : 0;
        }
    }
}

// Obstacle half-planes of every live RVO ego of the workgroup (RVOPolicy.py:56-57; the obstacle half of
// Agent::computeNewVelocity), on all lanes in five sub-steps: (1) one lane per (ego, rectangle) tests the four edges and
// appends the neighbours to the ego's candidate list (LDS counter); (2) one lane per ego turns its count into a share of the
// workgroup's dense (ego, candidate) work list; (3) one lane per (ego, candidate) finds the candidate's rank by counting -
// (squared distance, rectangle, edge) is the order Agent::insertObstacleNeighbor's insertion sort produces - and builds its
// half-plane (it does not depend on the earlier ones) into row `rank`; (4) one lane per (ego, candidate, earlier candidate)
// evaluates "already covered by that line" into a bit matrix; (5) one lane per ego walks its candidates in order: a candidate
// covered by a KEPT line is dropped (bit tests), the kept lines are compacted to rows 0 .. nobl-1 of its column and tested
// against the LP start.  Sub-steps 2 and 5 used to be an insertion sort and a nested loop per ego lane (27 000 of the phase's
// 38 000 cycles).  Needs: W.rect staged, W.lpc published, W.nobl zero, a barrier behind all three; every thread of the
// workgroup calls it (four barriers inside); the caller's next barrier publishes the result.  ko <= 32 (bit masks).
__device__ inline void obstacle_lines_phase3(const CagymDev& D, const Lds3& W, int M, int AS, int ko, int nagents, uint32_t inv_m) {
    const int tid = threadIdx.x, NTT = blockDim.x, Kobs = D.Kobs;
    float2* nbr = reinterpret_cast<float2*>(W.lp3);      // [ko][AS] candidates (squared distance, id) in the idle LP3 scratch
    int* todo = reinterpret_cast<int*>(nbr + ko * AS);   // [<= nagents * ko] (ego << 8 | candidate, later rank) behind them
    uint8_t* perm = reinterpret_cast<uint8_t*>(todo + ko * AS);  // [ko][AS] candidate at each rank
    uint32_t* cov = W.cov;                               // [ko][AS] bit s of entry (rank r, ego): line s covers candidate r
    const float inv_tho = 1.0f / 5.0f;
    OBSTAMP_BEGIN();
    for (int q = tid; q < nagents * Kobs; q += NTT) {
        const int a = q / Kobs, r = q - a * Kobs;
        if (!W.trvo[a]) continue;
        const int wl = (int)__umulhi((uint32_t)a, inv_m);
        if (r >= W.wnob[wl]) continue;
        const float px = (float)W.tpx[a], py = (float)W.tpy[a];
        const float radius = (float)((1 + 15e-2) * W.tr[a]), max_speed = (float)W.tpref[a];
        const float range = 5.0f * max_speed + radius, range_sq = range * range;
        const float4* rect = W.rect + (wl * Kobs + r) * 4;
        // the four edge tests first, then ONE atomic for the rectangle's neighbours (an atomic per edge was up to four dependent LDS
        // round trips per lane); the order of the entries does not matter: the ranking below sorts by (distance, id)
        float dq[4];
        int cntn = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dq[k] = -1.0f;
            float dsq;
            if (orca_edge_is_neighbour(rect, k, px, py, range_sq, dsq)) { dq[k] = dsq; cntn++; }
        }
        if (cntn > 0) {
            int slot = __hip_atomic_fetch_add(&W.nobl[a], cntn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (dq[k] >= 0.0f) {
                    if (slot < ko) nbr[slot * AS + a] = make_float2(dq[k], __int_as_float(4 * r + k));
                    slot++;
                }
        }
    }
    __syncthreads();
    WGTRACE(33);
    OBSTAMP(0);
    if (tid >= NTT - CAGYM_WAVE) {
        // the (ego, candidate) pairs of the whole workgroup as one dense list (the next sub-steps are rounds of full waves): every ego
        // is a lane of this wave, so its share is a prefix sum (50 lanes adding to one LDS counter cost 4 000 cycles)
        const int a = tid - (NTT - CAGYM_WAVE);
        int n = 0;
        if (a < nagents) {
            n = W.nobl[a];
            n = n < ko ? n : ko;
            W.nobl[a] = n;
        }
        int incl = n, mx = n;
        for (int off = 1; off < CAGYM_WAVE; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (a >= off) incl += v;
            const int u = __shfl_xor(mx, off);
            mx = u > mx ? u : mx;
        }
        const int base = incl - n;
        for (int i = 0; i < n; i++) todo[base + i] = (a << 8) | i;
        if (a == CAGYM_WAVE - 1) { W.flag[5] = incl; W.flag[6] = mx; }
    }
    __syncthreads();
    WGTRACE(34);
    OBSTAMP(9);
    const int ntodo = W.flag[5], nmax = W.flag[6];
    WGTRACE_VALUE(37, ntodo | (nmax << 16));
    for (int q = tid; q < ntodo; q += NTT) {
        const int a = todo[q] >> 8, i = todo[q] & 255;
        const int n = W.nobl[a];
        const float2 me = nbr[i * AS + a];
        int rank = 0;  // candidates before this one: smaller (squared distance, id)
        for (int l = 0; l < n; l++) {
            const float2 o = nbr[l * AS + a];
            rank += (o.x < me.x) || (o.x == me.x && __float_as_int(o.y) < __float_as_int(me.y));
        }
        const int wl = (int)__umulhi((uint32_t)a, inv_m);
        const float px = (float)W.tpx[a], py = (float)W.tpy[a], vx = (float)W.tvx[a], vy = (float)W.tvy[a];
        const float radius = (float)((1 + 15e-2) * W.tr[a]);
        float4 ln;
        const bool ok = orca_obstacle_line_of(W.rect + wl * Kobs * 4, __float_as_int(me.y), px, py, vx, vy, radius, inv_tho, ln);
        if (!ok) ln.z = __int_as_float(0x7fc00000);  // "this edge contributes no half-plane": NaN direction
        W.sorted[rank * AS + a] = ln;
        perm[rank * AS + a] = (uint8_t)i;
        cov[rank * AS + a] = 0u;
        todo[q] = (a << 8) | rank;  // (only this lane reads or writes entry q)
    }
    __syncthreads();
    WGTRACE(35);
    OBSTAMP(10);
    for (int w = tid; w < ntodo * nmax; w += NTT) {
        const int q = w / nmax, sidx = w - q * nmax;
        const int a = todo[q] >> 8, r = todo[q] & 255;
        if (sidx >= r) continue;
        const float4 ls = W.sorted[sidx * AS + a];
        if (ls.z != ls.z) continue;  // no half-plane at that rank: never kept, covers nothing
        const int wl = (int)__umulhi((uint32_t)a, inv_m);
        const int id = __float_as_int(nbr[perm[r * AS + a] * AS + a].y);
        const float px = (float)W.tpx[a], py = (float)W.tpy[a];
        const float radius = (float)((1 + 15e-2) * W.tr[a]);
        if (orca_edge_covered_by(W.rect + wl * Kobs * 4, id, px, py, radius, inv_tho, ls))
            __hip_atomic_fetch_or(&cov[r * AS + a], 1u << sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    WGTRACE(36);
    OBSTAMP(11);
    if (tid >= NTT - CAGYM_WAVE) {
        const int a = tid - (NTT - CAGYM_WAVE);
        if (a < nagents) {
            const int n = W.nobl[a];
            int nl = 0;
            if (n > 0) {
                const float2 s0 = W.lpc[a];
                bool viol = false;
                uint32_t kept = 0u;  // ranks whose line was kept
                for (int r = 0; r < n; r++) {
                    if (cov[r * AS + a] & kept) continue;  // already covered by a kept line
                    const float4 ln = W.sorted[r * AS + a];
                    if (ln.z != ln.z) continue;   // no half-plane from this edge
                    W.sorted[nl * AS + a] = ln;   // nl <= r: in place
                    nl++;
                    kept |= 1u << r;
                    viol |= detf(ln.z, ln.w, ln.x - s0.x, ln.y - s0.y) > 0.0f;
                }
                if (viol) W.busy[a] = 1;
            }
            W.nobl[a] = nl;
        }
    }
}

// What of the next ORCA solve depends on ego a alone: preferred velocity / LP start (the obstacle half-planes of worlds with
// rectangles follow in obstacle_lines_phase3, which counts on nobl = 0).
template <bool OBST>
__device__ __forceinline__ void ego_lp_inputs3(const CagymDev& D, const Lds3& W, int a, int M, int AS, int ko, uint32_t inv_m) {
    publish_pref_velocity3(W, a);
    if (OBST) W.nobl[a] = 0;
}

// unordered pair p of the workgroup -> (world of the workgroup, i, j); compile-time M or run-time M (magic division)
template <int MT>
__device__ __forceinline__ UPair upair_of(int p, int M) {
    if (MT > 0) return UnorderedPairs<MT>::of(p);
    UPair q;
    const uint32_t npw = (uint32_t)(M * (M - 1) / 2);
    const uint32_t inv_n = (uint32_t)(0x100000000ull / npw) + 1u, inv_m = (uint32_t)(0x100000000ull / (uint32_t)M) + 1u;
    q.wl = npw == 1u ? p : (int)__umulhi((uint32_t)p, inv_n);  // the magic number of divisor 1 does not fit 32 bits
    const int u = p - q.wl * (int)npw;
    const int H = (M - 1) / 2;
    if (u < M * H) {
        const int k = (int)__umulhi((uint32_t)u, inv_m);
        q.i = u - k * M;
        q.j = q.i + k + 1;
        if (q.j >= M) q.j -= M;
    } else {
        q.i = u - M * H;
        q.j = q.i + M / 2;
    }
    return q;
}

// order-preserving 64-bit key of a double (no NaN): unsigned order of the keys = order of the doubles
__device__ __forceinline__ unsigned long long gap_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
__device__ __forceinline__ double gap_of_key(unsigned long long k) {
    const unsigned long long b = k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull);
    return __longlong_as_double((long long)b);
}
#define CAGYM_GAP_INF 0xFFF0000000000000ull  /* gap_key(+inf) */

// ---- phase A body: pair distances of the moved state (env.py:630-655), OAS sort keys, fp32 squared distances ------------
// GAP: fold the pair's gap into the lower agent's running minimum (phase A only: S2 consumes and resets it; the prologue's and the
// reset path's calls rebuild keys / distances for the NEXT step's half-planes and rows and must leave it alone)
// DSQ: also the ORCA ranking keys (not in the POST half of a split step: the next step's half-planes are another launch's)
template <int MT, bool GAP, bool DSQ = true>
__device__ __forceinline__ void pair_distances3(const CagymDev& D, const Lds3& W, int p, int M, int MP) {
    const UPair q = upair_of<MT>(p, M);
    const int n = W.wn[q.wl];
    const int lo = q.wl * M + (q.i < q.j ? q.i : q.j), hi = q.wl * M + (q.i < q.j ? q.j : q.i);
    const int slo = lo - q.wl * M, shi = hi - q.wl * M;
    double klo = -INFINITY, khi = -INFINITY, gp = INFINITY;
    float dq = INFINITY;
    uint8_t ht = 0;
    if (shi < n) {  // slo < shi < n
        const double plx = W.tpx[lo], ply = W.tpy[lo], phx = W.tpx[hi], phy = W.tpy[hi];
        const double dx = phx - plx, dy = phy - ply;
        const double d = norm2(dx, dy);
        const double rl = W.tr[lo], rh = W.tr[hi];
        const bool skip = ST_POLICY(W.tst[hi]) == CAGYM_POL_STATIC && !D.collide_static;  // env.py:643 (Q8)
        const double cr = rl + rh;
        ht = (!skip && d <= cr) ? 1 : 0;
        if (!skip) gp = d - cr;  // lower index only (Q7)
        klo = d - rl - rh;
        khi = d - rh - rl;
        // Agent::computeNewVelocity's distSq of the pair, fp32: (float)p_other - (float)p_ego, squared (sign-symmetric)
        const float rpx = (float)phx - (float)plx, rpy = (float)phy - (float)ply;
        dq = rpx * rpx + rpy * rpy;
    }
    W.hit[lo * MP + shi] = ht;
    W.hit[hi * MP + slo] = ht;
    if (GAP && gp < INFINITY) atomicMin(&W.gmin[lo], gap_key(gp));  // (no lane waits for the result: ds_min_u64 without return)
    W.keys[lo * MP + shi] = klo;
    W.keys[hi * MP + slo] = khi;
    if (DSQ) {
        W.dsq[lo * MP + shi] = make_uint2((uint32_t)shi, __float_as_uint(dq));
        W.dsq[hi * MP + slo] = make_uint2((uint32_t)slo, __float_as_uint(dq));
    }
}

// the ORCA ranking keys of one unordered pair alone (pair_distances3 without the collision / observation half): what a one-step
// launch needs ahead of its own half-planes (its rows of step t - 1 do not exist)
template <int MT>
__device__ __forceinline__ void pair_dsq3(const Lds3& W, int p, int M, int MP) {
    const UPair q = upair_of<MT>(p, M);
    const int n = W.wn[q.wl];
    const int lo = q.wl * M + (q.i < q.j ? q.i : q.j), hi = q.wl * M + (q.i < q.j ? q.j : q.i);
    const int slo = lo - q.wl * M, shi = hi - q.wl * M;
    float dq = INFINITY;
    if (shi < n) {
        // Agent::computeNewVelocity's distSq of the pair, fp32: (float)p_other - (float)p_ego, squared (sign-symmetric)
        const float rpx = (float)W.tpx[hi] - (float)W.tpx[lo], rpy = (float)W.tpy[hi] - (float)W.tpy[lo];
        dq = rpx * rpx + rpy * rpy;
    }
    W.dsq[lo * MP + shi] = make_uint2((uint32_t)shi, __float_as_uint(dq));
    W.dsq[hi * MP + slo] = make_uint2((uint32_t)slo, __float_as_uint(dq));
}

// rank of slot sl in ego a's row of squared distances: nearest first, ties by lower index (Agent::insertAgentNeighbor).
// The row holds one 64-bit key per slot, (distance bits << 32) | slot: "nearer, or as near with a lower index" is ONE unsigned
// 64-bit comparison, and the count one add-with-carry per slot (two instructions per slot; the float version needed two
// comparisons, an index test, the logic between them and the add).  The own slot and the padding hold +inf.
template <int MT>
__device__ __forceinline__ int neighbour_rank3(const Lds3& W, int a, int sl, float dq, int MP) {
    const uint4* row = reinterpret_cast<const uint4*>(W.dsq + a * MP);  // MP is a multiple of 4: 16-byte aligned pairs of keys
    const unsigned long long me = ((unsigned long long)__float_as_uint(dq) << 32) | (unsigned)sl;
    int rank = 0;
    constexpr int MPT = MT > 0 ? ((MT + 1) & ~1) : 0;  // compile-time M: the keys beyond it are padding (+inf), never smaller
    if (MT > 0) {
#pragma unroll
        for (int l2 = 0; l2 < MPT; l2 += 2) {
            const uint4 v = row[l2 >> 1];
            rank += ((((unsigned long long)v.y << 32) | v.x) < me) ? 1 : 0;
            rank += ((((unsigned long long)v.w << 32) | v.z) < me) ? 1 : 0;
        }
    } else {
        for (int l2 = 0; l2 < MP; l2 += 2) {
            const uint4 v = row[l2 >> 1];
            rank += ((((unsigned long long)v.y << 32) | v.x) < me) ? 1 : 0;
            rank += ((((unsigned long long)v.w << 32) | v.z) < me) ? 1 : 0;
        }
    }
    return rank;
}

// ---- phase B body: ORCA half-planes of one unordered pair, ranked into both egos' nearest-first line lists -----------------
// (Round 3 measured storing them unsorted and letting the LP group of a busy ego rank them - "lazy ranking", together with the
// all-linearProgram1-results-up-front solver: slower as a pair; tools/lp_upfront/ holds both and the numbers.)
template <int MT>
__device__ __forceinline__ void half_planes3(const CagymDev& D, const Lds3& W, int p, int M, int MP, int AS, int ko) {
    const UPair q = upair_of<MT>(p, M);
    const int n = W.wn[q.wl];
    const int a = q.wl * M + q.i, b = q.wl * M + q.j;
    const bool both = q.i < n && q.j < n;
    const bool on_a = both && W.trvo[a] != 0;
    const bool on_b = both && W.trvo[b] != 0;
    if (!(on_a || on_b)) return;
    const float vax = (float)W.tvx[a], vay = (float)W.tvy[a];
    const OrcaPair g = orca_pair((float)W.tpx[a], (float)W.tpy[a], vax, vay, (float)((1 + 15e-2) * W.tr[a]), (float)D.dt,
                                 W.tpx[b], W.tpy[b], W.tvx[b], W.tvy[b], W.tr[b]);
    PMARK("hp_pair_done");
    if (on_a) {
        const float c = W.tcoop[a];
        const float4 ln = make_float4(vax + c * g.ux, vay + c * g.uy, g.zx, g.zy);
        const float2 s0 = W.lpc[a];
        if (detf(ln.z, ln.w, ln.x - s0.x, ln.y - s0.y) > 0.0f) W.busy[a] = 1;
        const int rank = neighbour_rank3<MT>(W, a, q.j, g.d2, MP);
        if (rank < D.maxnb) W.sorted[(ko + rank) * AS + a] = ln;
    }
    if (on_b) {
        const float c = W.tcoop[b];
        const float4 ln = make_float4((float)W.tvx[b] - c * g.ux, (float)W.tvy[b] - c * g.uy, -g.zx, -g.zy);
        const float2 s0 = W.lpc[b];
        if (detf(ln.z, ln.w, ln.x - s0.x, ln.y - s0.y) > 0.0f) W.busy[b] = 1;
        const int rank = neighbour_rank3<MT>(W, b, q.i, g.d2, MP);
        if (rank < D.maxnb) W.sorted[(ko + rank) * AS + b] = ln;
    }
}

// ---- one 64-row chunk of the OtherAgentsStates table (sensors/OtherAgentsStatesSensor.py:11-77), straight to HBM ----------
__device__ __forceinline__ void oas_row3(const Lds3& W, float* oas_out, int p, int npairs, int M, int MP, int K, int wpw,
                                         int worlds_valid, uint32_t inv_m) {
    PMARK("oas_begin");
    if (p >= npairs) return;
    const PairIdx q = pair_of(p, M, inv_m);
    if (q.j == q.sl || q.wl >= worlds_valid) return;
    const int n = W.wn[q.wl];
    float* my = oas_out + ((size_t)blockIdx.x * wpw * M + q.a) * K * 10;
    const double kj = W.keys[q.a * MP + q.j];
    if (!(kj > -INFINITY)) {  // unused row: zero (rows n-1 .. K-1 of an active agent, every row of an empty slot)
        // (its own stores: merged with the live rows' values through a common array, the zeros cost ~20 register moves per chunk)
        const int row = q.sl < n ? q.j - 1 : (q.j < q.sl ? q.j : q.j - 1);
        float2* r2 = reinterpret_cast<float2*>(my + row * 10);
        const float2 z = make_float2(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 5; c++) r2[c] = z;
        return;
    }
    int before = 0;  // descending key, ties by descending index (stable sort, reversed: :28-34)
    const double2* krow = reinterpret_cast<const double2*>(W.keys + q.a * MP);
    for (int l2 = 0; l2 < M; l2 += 2) {  // keys beyond the world's slots (and the pad of an odd M) are -inf
        const double2 kk = krow[l2 >> 1];
        before += (kk.x > kj) || (kk.x == kj && l2 + 0 > q.j);
        before += (kk.y > kj) || (kk.y == kj && l2 + 1 > q.j);
    }
    const int row = before;
    PMARK("oas_ranked");
    const int b = q.a - q.sl + q.j;
    const double dx = W.tpx[b] - W.tpx[q.a], dy = W.tpy[b] - W.tpy[q.a];
    const double prx = W.tprx[q.a], pry = W.tpry[q.a], orx = -pry, ory = prx;
    const double ovx = W.tvx[b], ovy = W.tvy[b], orad = W.tr[b];
    float v[10];
    v[0] = (float)dx;
    v[1] = (float)dy;
    v[2] = (float)dot2(dx, dy, prx, pry);
    v[3] = (float)dot2(dx, dy, orx, ory);
    v[4] = (float)dot2(ovx, ovy, prx, pry);
    v[5] = (float)dot2(ovx, ovy, orx, ory);
    v[6] = (float)orad;
    v[7] = (float)(W.tr[q.a] + orad);
    v[8] = (float)kj;
    v[9] = ST_POLICY(W.tst[b]) == CAGYM_POL_STATIC ? 1.f : 2.f;
    PMARK("oas_store");
    float2* r2 = reinterpret_cast<float2*>(my + row * 10);  // rows are 40 B: 8-byte aligned
#pragma unroll
    for (int c = 0; c < 5; c++) r2[c] = make_float2(v[2 * c], v[2 * c + 1]);
}

// scalar observation keys of agent slot a of the workgroup (agent.py:244-248, config.py:104-215) + n_observed
__device__ __forceinline__ void ego_obs3(const CagymDev& D, const Lds3& W, float* ego_out, int a, int M, int wpw, uint32_t inv_m) {
    if (a >= wpw * M) return;
    const int wl = (int)__umulhi((uint32_t)a, inv_m);
    const int slot = a - wl * M;
    const int world = blockIdx.x * wpw + wl;
    if (world >= D.N) return;
    const int n = W.wn[wl];
    const bool active = slot < n;
    const int nobs = active ? n - 1 : 0;
    const size_t aidx = (size_t)world * M + slot;
    D.n_observed[aidx] = nobs;
    if (!ego_out) return;
    float4* e = reinterpret_cast<float4*>(ego_out + aidx * CAGYM_EGO_WIDTH);
    if (active) {
        const double px = W.tpx[a], py = W.tpy[a];
        e[0] = make_float4((float)W.tdg[a], (float)(W.tgx[a] - px), (float)(W.tgy[a] - py), (float)W.tr[a]);
        e[1] = make_float4((float)W.the[a], (float)W.th[a], (float)px, (float)py);
        e[2] = make_float4((float)W.tpref[a], (float)nobs, ST_POLICY(W.tst[a]) == CAGYM_POL_LEARNING ? 1.f : 0.f, 0.f);
    } else {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);  // (a chained assignment reads e[1] back from memory)
        e[0] = z; e[1] = z; e[2] = z;
    }
}

// LaserScanSensor.sense restricted to the samples klo..khi of beam b (every sample outside can be shown not to hit): the running
// hit count of sensors/LaserScanSensor.py:45-58 only changes at hits, so "the last sample whose count is 1" is sample 15 when
// exactly one hit lies in the interval, the sample before the second hit when there are more, none without a hit.
// world_to_cell for the one-step kernel's beam samples: on its usual path (|q| < 1e6, not within 1e-7 of a cell border) the floors fit
// an int as they are - the clamps of world_to_cell are the identity there (12 of a sample's ~60 instructions); every other
// input takes world_to_cell itself.  (In world_to_cell for everybody the extra branch cost the OBST roll-out kernels 8 spilled registers.)
template <bool LEAN>
__device__ __forceinline__ bool sample_cell(double x, double y, int& gx, int& gy) {
    if (!LEAN) return world_to_cell(x, y, gx, gy);
    const double qy = y * 10.0, qx = x * 10.0;
    const bool safe = fabs(qy - rint(qy)) > 1e-7 && fabs(qx - rint(qx)) > 1e-7 && fabs(qy) < 1e6 && fabs(qx) < 1e6;
    if (!safe) return world_to_cell(x, y, gx, gy);
    gx = (int)floor((30 / 2.) / 0.1 - qy);
    gy = (int)floor((30 / 2.) / 0.1 + qx);
    return gx >= 0 && gy >= 0 && gx < CAGYM_MAPD && gy < CAGYM_MAPD;
}
#ifndef CAGYM_LASER_BATCH_ROLLOUT
#define CAGYM_LASER_BATCH_ROLLOUT 4  /* gathers in flight per sampled beam in the roll-out kernels (register limit); the one-step kernel keeps 8 */
#endif
template <int BATCH>
__device__ __forceinline__ float laserscan_beam_range(const uint32_t* map, double px, double py, double h, double radius, int b,
                                                      int klo, int khi) {
    int egx, egy;
    const bool ego_in = world_to_cell(px, py, egx, egy);
    const double rr = radius / 0.1, r2 = rr * rr;
    // the own-disc mask (dx^2 + dy^2 < r2 on cell offsets) on integers: for an integer d, d < r2 <=> d <= ceil(r2) - 1 (the sixteen
    // samples of a beam are ~1000 instructions per round of 64 beams, conversions and fp64 from end to end: what is taken out shows)
    const int disc_lim = r2 < 1e9 ? (int)ceil(r2) - 1 : 0x7fffffff;
    const double astep = (kPi - (-kPi)) / 15.0, rstep = 2 * kPi / 16;
    const double ang0 = b == 15 ? kPi : (double)b * astep + (-kPi);
    double sa, ca;
    sincos(ang0 + h, &sa, &ca);
    // the raster words are requested BATCH at a time before the first one is looked at (the raster is L2-resident and a gather's
    // latency is ~1000 cycles: sample after sample, a wave's 44 listed beams took 7.5 us; all sixteen at once cost the OBST roll-out
    // kernels their registers); the hits of the interval become a 16-bit mask
    uint32_t hits = 0u;
#pragma unroll
    for (int half = 0; half < 16 / BATCH; half++) {
        uint32_t word[BATCH];
        int sh[BATCH];
        uint32_t valid = 0u;
#pragma unroll
        for (int q = 0; q < BATCH; q++) {
            const int k = half * BATCH + q;
            bool in = k >= klo && k <= khi;
            int gx = 0, gy = 0;
            if (in) {
                const double rg = 0.0 + (double)k * rstep;
                const double x = px + rg * ca, y = py + rg * sa;
                in = sample_cell<(BATCH >= 8)>(x, y, gx, gy);
                if (in && ego_in) {
                    const int dx = gy - egy, dy = gx - egx;  // both cells lie in the 300 x 300 map: no overflow
                    in = !(dx * dx + dy * dy <= disc_lim);
                }
            }
            valid |= in ? (1u << q) : 0u;
            sh[q] = gy & 31;
            word[q] = map[in ? gx * CAGYM_MAPW + (gy >> 5) : 0];  // unconditional, clamped address
        }
#pragma unroll
        for (int q = 0; q < BATCH; q++) hits |= (((word[q] >> sh[q]) & (valid >> q)) & 1u) << (half * BATCH + q);
    }
    // "the last sample whose running hit count is 1": the sample before the second hit, or the beam's last sample when only one hits
    int count = hits ? 1 : 0, last = -1;
    if (hits) {
        const uint32_t rest = hits & (hits - 1u);
        last = rest ? (int)__builtin_ctz(rest) - 1 : 15;
        count = rest ? 2 : 1;
    }
    if (count == 1) last = 15;  // no further hit beyond the interval: the count stays 1 to the end of the beam
    const double range = last >= 0 ? 0.0 + (double)last * rstep : 6.0;
    return (float)(1 - range / 6);
}

// LaserScan of every agent slot of the workgroup, by whichever waves call it (each at its own time; no barrier inside).  An occupied
// raster cell reaches at most one cell (0.1 m) beyond its rectangle, so a beam whose segment misses every rectangle of the world
// inflated by 0.25 m reads 0.0 (= 1 - 6/6) without looking at the raster, and a beam that meets some only needs the samples inside
// those crossings.  Part 1, claimed in wave-passes of 4 agents x 16 beams: fp32 slab test of the beam against the staged rectangles;
// the beams that need the raster go to the pass's 64 entries of ONE list of the workgroup.  Part 2, once every pass is in (LDS counter,
// release / acquire): the fp64 sampling in rounds of 64 LISTED beams - a round costs the same ~1 000 instructions whether 20 or 64 of
// its lanes hold a beam, so the rounds must be dense: per-wave lists (round 2, 16 agents each) needed 5 rounds per workgroup where
// 150 listed beams fill 3 (cfg4: B phase 14.4 -> x us, tools/cfg4_timeline.py).  A rectangle that leaves the map (numpy's negative-index
// wrap puts its cells elsewhere) or unstaged rectangles (no RVO agent among them: ko == 0) disable the shortcut: every beam of the
// world is sampled in full.  Counters W.flag[7..10] must be zero on entry of the first wave (publish step, reset path).
template <int BATCH>
__device__ __forceinline__ void laser_scan3(const CagymDev& D, const Lds3& W, float* laser_out, int M, int wpw, int AS, uint32_t inv_m, int ko) {
    const int lane = threadIdx.x & (CAGYM_WAVE - 1);
    uint16_t* lst = W.blist;                                      // (agent slot << 4) | beam
    uint8_t* rng = reinterpret_cast<uint8_t*>(W.blist + AS * 16);  // (first sample << 4) | last sample
    const float rstep = (float)(2 * kPi / 16), reach = 15.0f * rstep + 0.05f, infl = 0.25f;
    uint8_t* cnt = rng + AS * 16;                                  // [passes] beams listed by each pass (pass c owns entries c * 64 ..)
    const int nag = wpw * M, npass = (nag + 3) / 4;
    WGTRACE_W1(29);
    // (ONE lane-0 region per iteration, at its top: "the pass I just finished is in" (release: a wave's LDS operations complete in
    // order) and the next claim.  With a second `if (lane == 0)` atomic at the END of the body the compiled loop gained an inner
    // loop between the claim and the body that peeled lane 0 off and re-entered the body with claim 0 for the other lanes (visible
    // in the ISA), and the wait below never ended.)
    bool finished_one = false;
    for (;;) {
        int c = 0;
        if (lane == 0) {
            if (finished_one) __hip_atomic_fetch_add(&W.flag[8], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            c = __hip_atomic_fetch_add(&W.flag[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        c = __builtin_amdgcn_readlane(c, 0);  // lane 0's claim whatever the loop's shape made of the exec mask (not "the first active lane")
        if (c >= npass) break;
        finished_one = true;
        const int a = c * 4 + (lane >> 4), b = lane & 15;
        bool need = false;
        int klo = 0, khi = 15;
        if (a < nag) {
            const int wl = (int)__umulhi((uint32_t)a, inv_m);
            const int slot = a - wl * M;
            const int world = blockIdx.x * wpw + wl;
            if (world < D.N) {
                if (slot < W.wn[wl]) {
                    if (ko <= 0) {
                        need = D.map_bits && D.sc_nobst[W.wsc[wl]] > 0;
                    } else if (W.wnob[wl] > 0) {
                        const double astep = (kPi - (-kPi)) / 15.0;
                        const double ang0 = b == 15 ? kPi : (double)b * astep + (-kPi);
                        float sa, ca;
                        __sincosf((float)(ang0 + W.th[a]), &sa, &ca);
                        const float px = (float)W.tpx[a], py = (float)W.tpy[a];
                        const float icx = 1.0f / ca, icy = 1.0f / sa;
                        const float pxl = px + infl, pxu = px - infl, pyl = py + infl, pyu = py - infl;  // (the test is conservative by 0.15 m: roundings do not matter)
                        float tlo = INFINITY, thi = -INFINITY;
                        const float4* R = W.rect + (size_t)wl * D.Kobs * 4;
                        for (int k = 0; k < W.wnob[wl]; k++) {
                            const float4 r = R[4 * k];
                            if (R[4 * k + 3].y != 1.0f) { tlo = 0.f; thi = reach; continue; }  // leaves the map: no shortcut
                            const float tx1 = (r.x - pxl) * icx, tx2 = (r.z - pxu) * icx;
                            const float ty1 = (r.y - pyl) * icy, ty2 = (r.w - pyu) * icy;
                            const float t1 = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), 0.f);
                            const float t2 = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), reach);
                            if (t1 <= t2) { tlo = fminf(tlo, t1); thi = fmaxf(thi, t2); }
                        }
                        need = tlo <= thi;
                        if (need) {
                            klo = (int)floorf((tlo - 0.05f) / rstep);
                            khi = (int)ceilf((thi + 0.05f) / rstep);
                            klo = klo < 0 ? 0 : klo;
                            khi = khi > 15 ? 15 : khi;
                        }
                    }
                }
                if (!need) laser_out[((size_t)world * M + slot) * 16 + b] = 0.f;  // inactive slot, empty map, or no rectangle in reach
            }
        }
        const unsigned long long m = __ballot(need);
        if (need) {
            const int i = c * CAGYM_WAVE + __popcll(m & ((1ull << lane) - 1ull));
            lst[i] = (uint16_t)((a << 4) | b);
            rng[i] = (uint8_t)((klo << 4) | khi);
        }
        cnt[c] = (uint8_t)__popcll(m);  // (every lane stores the same byte: no branch)
    }
    WGTRACE_W1(30);
    // every pass was claimed by some wave that is running it now (nobody waits for a wave that never calls); the wait is bounded all
    // the same (cagym_spin.h): a count that cannot arrive becomes CAGYM_E_DEVICE at the next entry point, not a hung GPU
    if (!lds_wait_ge(&W.flag[8], npass)) *D.dev_status = CAGYM_DEVERR_LASER_WAIT;
    int n = 0;
    for (int q = 0; q < npass; q++) n += cnt[q];
    WGTRACE_W1_VALUE(32, n);
    for (;;) {
        int r = 0;
        if (lane == 0) r = __hip_atomic_fetch_add(&W.flag[10], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        r = __builtin_amdgcn_readlane(r, 0);
        if (r * CAGYM_WAVE >= n) break;
        int i = r * CAGYM_WAVE + lane;  // index into the concatenation of the passes' lists
        if (i < n) {
            int q = 0;
            for (; q < npass; q++) {  // (at most 16 passes; the counts are broadcast reads)
                const int cq = cnt[q];
                if (i < cq) break;
                i -= cq;
            }
            const int e = lst[q * CAGYM_WAVE + i], kk = rng[q * CAGYM_WAVE + i];
            const int a = e >> 4, b = e & 15;
            const int wl = (int)__umulhi((uint32_t)a, inv_m);
            const int slot = a - wl * M;
            const int world = blockIdx.x * wpw + wl;
            const uint32_t* map = D.map_bits + (size_t)W.wsc[wl] * CAGYM_MAPD * CAGYM_MAPW;
            laser_out[((size_t)world * M + slot) * 16 + b] = laserscan_beam_range<BATCH>(map, W.tpx[a], W.tpy[a], W.th[a], W.tr[a], b, kk >> 4, kk & 15);
        }
    }
    WGTRACE_W1(31);
}

// claim-and-process loop of the observation workers: chunks 0 .. nck-1 are 64 directed pairs each, chunk nck is the
// scalar-observation store of the agent slots; the LaserScan (when asked for) follows behind them (laser_scan3 has its own
// claim counters).  Every wave of the workgroup may call it; a wave leaves the chunk loop when the counter has run past the
// last chunk (every wave reaches that: the counter only grows).
template <bool OBST, int BATCH = 8>
__device__ __forceinline__ void observation_chunks3(const CagymDev& D, const Lds3& W, const CagymOut& o, int npairs, int M, int MP,
                                                    int K, int wpw, int worlds_valid, uint32_t inv_m, int ko = 0, int AS = 0) {
    const int lane = threadIdx.x & (CAGYM_WAVE - 1);
    const int nck = (npairs + CAGYM_WAVE - 1) / CAGYM_WAVE;
    for (;;) {
        int c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(&W.flag[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        c = __builtin_amdgcn_readlane(c, 0);
        if (c > nck) break;
        if (c < nck) {
            if (o.obs_oas) oas_row3(W, o.obs_oas, c * CAGYM_WAVE + lane, npairs, M, MP, K, wpw, worlds_valid, inv_m);
        } else {
            ego_obs3(D, W, o.obs_ego, lane, M, wpw, inv_m);
        }
    }
    if (OBST && o.laserscan) laser_scan3<BATCH>(D, W, o.laserscan, M, wpw, AS, inv_m, ko);
}

__device__ __forceinline__ CagymOut out_slice3(const CagymOut& out, int t, size_t N, size_t NM, int M) {
    CagymOut o;
    o.obs_oas = out.obs_oas ? out.obs_oas + (size_t)t * NM * (M - 1) * 10 : nullptr;
    o.obs_ego = out.obs_ego ? out.obs_ego + (size_t)t * NM * CAGYM_EGO_WIDTH : nullptr;
    o.laserscan = out.laserscan ? out.laserscan + (size_t)t * NM * 16 : nullptr;
    o.reward = out.reward ? out.reward + (size_t)t * NM : nullptr;
    o.flags = out.flags ? out.flags + (size_t)t * NM : nullptr;
    o.game_over = out.game_over ? out.game_over + (size_t)t * N : nullptr;
    return o;
}

// n_steps env.step() calls of the workgroup's worlds.  ext: external actions of the (single) step or null.
// OBST: the handle's worlds may hold rectangles (max_obstacles > 0): wall test, LaserScan, and - when RVO agents live among
// them (D.ko > 0) - obstacle half-planes with LP groups of 4 half-planes per lane.  The free-space instantiation carries
// none of that code (it cost the headline kernel 7 VGPRs and a spill).
// ONE: the one-step launch (k_step3: n_steps == 1 at compile time)
// POST: the second half of a split step (k_step_post3, needs ONE and any_rvo == false): the RVO egos' new velocities were solved by
// k_step_pre3 (D.lp_vel); no half-plane rows / LP scratch / neighbour keys in LDS (carve_lds3_post)
template <int NT, int MT, int WPWT, bool AUTO_RESET, bool OBST, bool ONE = false, bool POST = false>
__device__ inline void run_steps3(const CagymDev& D, unsigned char* smem, const float* ext, const CagymOut& out, int n_steps,
                                  bool any_rvo) {
    constexpr int NWAVES = NT / CAGYM_WAVE;
    constexpr int GW = cagym_gw3(MT), NG = NT / GW, NGW = CAGYM_WAVE / GW;
    constexpr bool TWO = cagym_two3(MT);  // more than GW + 1 half-planes possible
    constexpr int NL = TWO ? 2 * GW : GW + 1;  // compile-time bound on the half-planes of one ego (free space)
    const int M = MT ? MT : D.M, K = M - 1, MP = cagym_mp(M);
    const int AS = cagym_as(M, WPWT);
    const int ko = OBST ? D.ko : 0;  // 2 * Kobs: an agent outside a rectangle sees at most 2 of its edges from their right side
    constexpr int LPL = cagym_lpl3(MT, OBST);
    constexpr bool OVL = ONE && OBST && !POST;  // the one-step launch among rectangles: early and late arrays share LDS bytes (carve_lds3_ovl)
    const Lds3 W = POST ? carve_lds3_post(smem, M, AS, ko / 2) : OVL ? carve_lds3_ovl(smem, M, AS, NT, ko) : carve_lds3(smem, M, AS, NT, ko, LPL, OBST, MT);
    // the neighbour keys live in the LP scratch (a one-step launch builds no half-planes for a next step: no keys after its prologue)
    const bool dsq_shared = !ONE && (OBST ? cagym_obst_alias(M, AS, NT, ko, LPL) : cagym_dsq_aliased(false, MT));
    LaneCtx C = make_ctx2(D, M, WPWT ? WPWT : CAGYM_WAVE / M);
    const uint32_t inv_m = (uint32_t)(0x100000000ull / (uint32_t)M) + 1u;
    const int nagents = C.wpw * M;               // agent slots of this workgroup (<= 64: wave 0)
    const int npairs = C.wpw * M * M;            // directed pair slots
    const int nup = C.wpw * (M * (M - 1) / 2);   // unordered pairs
    const size_t NM = (size_t)D.N * M;
    float ep_ret = 0.f;
    int ep_len = 0;
    STAMP_BEGIN();
    // ---- prologue: agent records -> LDS; the LP inputs of the first step ---------------------------------------------------
    {
        const int tid = threadIdx.x;
        const bool agent_lane = tid < nagents;
        if (!agent_lane) C.valid = C.active = false;
        if (agent_lane) {
            Agent A = {};
            if (C.valid) {
                load_agent(D, A, (size_t)C.world * M + C.slot);
                if (C.slot == 0) { ep_ret = D.ep_return[C.world]; ep_len = D.ep_len[C.world]; }
            }
            lds3_store_agent(W, A, tid);
            W.tmoved[tid] = 0;
            W.trvo[tid] = live_rvo(A.st, C.valid && C.active);
            W.nobl[tid] = 0;
            if (!OVL) W.gmin[tid] = CAGYM_GAP_INF;  // (OVL: the late arrays are initialised behind the linear programs, whose rows they overlay)
            if (C.wl < C.wpw && C.slot == 0) {
                W.wn[C.wl] = C.valid ? C.n : 0;
                W.wsc[C.wl] = C.valid ? (int)(((long long)C.world + (long long)C.episode * D.N) % D.S) : 0;
            }
            // constant entries of the distance / key rows: own slot and the padding
            if (!POST && !OVL) W.dsq[tid * MP + C.slot] = make_uint2((uint32_t)C.slot, 0x7f800000u);
            if (!OVL) {
                W.hit[tid * MP + C.slot] = 0;
                W.keys[tid * MP + C.slot] = -INFINITY;
            }
            for (int l = M; l < MP; l++) {
                if (!POST && !OVL) W.dsq[tid * MP + l] = make_uint2((uint32_t)l, 0x7f800000u);
                if (!OVL) {
                    W.hit[tid * MP + l] = 0;
                    W.keys[tid * MP + l] = -INFINITY;
                }
            }
        }
        if (tid < 16) W.flag[tid] = 0;
        __syncthreads();
        WGTRACE1(20);
        if (OBST && ko > 0) {
            stage_rects3(D, W, C.wpw, C.worlds_valid);
            __syncthreads();
        }
        WGTRACE1(21);
        if (agent_lane && any_rvo) ego_lp_inputs3<OBST>(D, W, tid, M, AS, ko, inv_m);
        else if (agent_lane) publish_pref_velocity3(W, tid);
        if (OBST && ko > 0 && any_rvo) {
            __syncthreads();
            obstacle_lines_phase3(D, W, M, AS, ko, nagents, inv_m);
        }
        WGTRACE1(22);
        if (any_rvo) {
            if (OVL && agent_lane) {
                // the neighbour keys take the bytes of the candidate lists: those are dead once every lane has passed the last barrier
                // inside obstacle_lines_phase3 (its last sub-step, on the last wave, reads the coverage bits and the rows only)
                W.dsq[tid * MP + C.slot] = make_uint2((uint32_t)C.slot, 0x7f800000u);
                for (int l = M; l < MP; l++) W.dsq[tid * MP + l] = make_uint2((uint32_t)l, 0x7f800000u);
            }
            for (int p = tid; p < nup; p += NT) {
                if (ONE) pair_dsq3<MT>(W, p, M, MP);  // (no rows of a step t - 1 to serve: the OAS keys / collision bits come with phase A)
                else pair_distances3<MT, false>(D, W, p, M, MP);
            }
            __syncthreads();
            WGTRACE1(23);
            for (int p = tid; p < nup; p += NT) half_planes3<MT>(D, W, p, M, MP, AS, ko);
        }
        __syncthreads();
    }
    WGTRACE(1);
#pragma nounroll
    for (int t = 0; t < n_steps; t++) {
        int tid = threadIdx.x;
        // opaque per step: keeps the compiler from hoisting every index derived from tid out of the step loop
        asm volatile("" : "+v"(tid));
        const int wave = tid >> 6;
        const bool agent_lane = tid < nagents;
        // the flat agent index as an opaque 32-bit value per step: as a loop invariant the compiler keeps several 64-bit forms of it
        // (x 1, x 4, + base) alive across the whole loop - in the kernels held to 128 VGPRs they went to scratch memory and came back
        // in S2 one round trip per output store (N * M < 2^31: cagym_create)
        int aidx32 = C.world * M + C.slot;
        asm volatile("" : "+v"(aidx32));
        const size_t aidx = (size_t)(unsigned)aidx32;
        const CagymOut o_prev = out_slice3(out, t > 0 ? t - 1 : 0, (size_t)D.N, NM, M);  // rows of step t-1 (used when t > 0)
        const CagymOut o = out_slice3(out, t, (size_t)D.N, NM, M);
        const bool lagging = __builtin_amdgcn_readfirstlane(W.flag[4]) != 0;  // some ego needed linearProgram3 in the previous step
        prio_chain3(lagging);
        WAVETRACE(t, 0);
        PMARK("C_begin");
        // ---- phase C: linearProgram2/3 of every busy ego on a GW-lane group (first waves) ------------------------------
        int lp_waves = 0;
        if (any_rvo) {
            int cnt;
            {
                const int lane = tid & (CAGYM_WAVE - 1);
                const bool fl = lane < nagents && W.busy[lane] != 0;
                const unsigned long long bm = __ballot(fl);
                cnt = __popcll(bm);
                if (fl) W.lpk[__popcll(bm & ((1ull << lane) - 1ull))] = lane;
            }
            WGTRACE_BUSY(cnt);
            lp_waves = (cnt + NGW - 1) / NGW;
            if (lp_waves > NWAVES) lp_waves = NWAVES;
            const int g = tid / GW, j = tid & (GW - 1);
            bool worked = false;
            STAMP(8);  // busy list
            WAVETRACE(t, 1);
            LPCOUNT_DBG_DECL();
            PMARK("C_lp_loop");
            // (cfg4: nearly every ego among rectangles is busy - median 35 of a workgroup's 36 RVO agents - so 70 % of the workgroups run
            // a second round of groups for a handful of egos.  Groups of FOUR lanes with four half-planes per lane - one round for up
            // to 64 egos, orca_lp_group_n<4, 4> - were measured slower: this phase 8.6 -> 9.4 us, tools/cfg4_timeline.py.  So was
            // letting the waves CLAIM the batches beyond the first round from an LDS counter, whichever finishes first: the second
            // round left wave 0 - its stamp 8.7 -> 6.1 us - but the workgroup's duration stayed at 45 us and the free-space roll-out
            // kernel picked up 12 bytes of scratch from the changed loop.)
            for (int base = 0; base < cnt; base += NG) {
                const int idx = base + g;
                if (idx < cnt) {
                    worked = true;
                    const int a = W.lpk[idx];
                    const int wl = (int)__umulhi((uint32_t)a, inv_m);
                    const int n = W.wn[wl];
                    const int nn = (n - 1) < D.maxnb ? (n - 1) : D.maxnb;
                    const float2 pv = W.lpv[a];
                    const float rad = W.lpr[a];
                    float vx, vy;
                    if (OBST) {
                        // two half-planes per lane serve 2 GW lines; only a wave that holds an ego with more runs the 4-per-lane code
                        const int nol = W.nobl[a];
                        float4* P = W.lp3 + (OVL ? 2 : LPL) * (tid & ~(GW - 1));  // (OVL: projected agent lines only, PC)
                        if (__ballot(nol + nn > 2 * GW) != 0ull)
                            orca_lp_group_n<GW, LPL, OVL>(W.sorted, P, a, j, nol, nn, ko, rad, pv.x, pv.y, vx, vy, AS, &W.flag[3]);
                        else
                            orca_lp_group_n<GW, 2, OVL>(W.sorted, P, a, j, nol, nn, ko, rad, pv.x, pv.y, vx, vy, AS, &W.flag[3]);
                    } else {
                        orca_lp_group<GW, TWO, (MT > 0 ? MT - 1 : 0)>(W.sorted, W.lp3 + LPL * (tid & ~(GW - 1)), a, j, nn, rad, pv.x, pv.y, vx, vy, AS, &W.flag[3],
                                                                      LPCOUNT_DBG(), LPWT_ROWS(t, base));
                    }
                    if (j == 0) W.lpc[a] = make_float2(vx, vy);
                }
            }
            LPCOUNT_FOLD(wave, tid, cnt);  // (diagnostic) lockstep trip counts of wave 0: the longest group sets the wave's time
            WAVETRACE(t, 2);
            PMARK("C_lp_done");
            // a wave that solved programs publishes them: LDS operations of one wave complete in order, the release
            // makes the compiler keep that order
            if (__builtin_amdgcn_readfirstlane((int)(__ballot(worked) != 0ull)) && (tid & (CAGYM_WAVE - 1)) == 0)
                __hip_atomic_fetch_add(&W.flag[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        STAMP(1);
        WGTRACE1(24);
        PMARK("D_begin");
        // ---- phase D: wave 0 = S1 (_take_action, env.py:287-340) in registers; the other waves = OAS rows of step t-1 ------
        //      (Round 3 measured two alternatives, both bit-identical and neither kept: S1 stored straight into a second copy of
        //      the S1 fields, one barrier instead of "barrier, publish, barrier": 3.97 vs 3.92 ms per 512-step launch on the same
        //      box, VGPRs 128 -> 101; and, on top of that, S1 run by each LP wave right behind its own programs plus a "quiet agents"
        //      pass on an idle wave: three executions of the S1 instruction stream per workgroup-step, wave-VALU instructions per
        //      workgroup-step 3 400 -> 4 205, 4.68 ms - the slowest LP wave sets LP + S1 either way.  profiles/r3/ab_pingpong_s1.txt)
        Agent A;
        bool moved = false;
        const bool s1_lane = agent_lane && C.valid && C.active;
        if (wave == 0) {
            // wait for the other LP waves (every LP wave increments exactly once per step; bounded all the same, cagym_spin.h)
            if (lp_waves > 0 && !lds_wait_ge(&W.flag[2], lp_waves)) *D.dev_status = CAGYM_DEVERR_LP_WAIT;
            STAMP(2);
            WAVETRACE(t, 3);
            PMARK("D_s1_begin");
            if (s1_lane) {
                A = lds3_load_agent(W, tid);
                float a0 = 0.f, a1 = 0.f;
                HeadingHint hint;
                hint.valid = false;
                if (!(A.st & CAGYM_FLAG_DONE)) {
                    double d0 = 0.0, d1 = 0.0;
                    switch (ST_POLICY(A.st)) {
                        case CAGYM_POL_STATIC: break;
                        case CAGYM_POL_NONCOOP: d0 = A.pref; d1 = -A.he; break;
                        case CAGYM_POL_EXTERNAL: case CAGYM_POL_IGMCTS: case CAGYM_POL_GA3C:
                            if (ext) { d0 = (double)ext[2 * aidx]; d1 = (double)ext[2 * aidx + 1]; }
                            break;
                        case CAGYM_POL_LEARNING:
                            if (ext) { d1 = 4.0 * (2. * (double)ext[2 * aidx + 1] - 1.); d0 = A.pref * (double)ext[2 * aidx]; }
                            else { d1 = -4.0; }
                            break;
                        case CAGYM_POL_CARRL: d0 = 1.0; d1 = carrl_heading(ext ? (int)ext[2 * aidx] : 0); break;
                        case CAGYM_POL_RVO: {
                            // LP result, or the clipped preferred velocity of an ego that needed none (split step: from k_step_pre3)
                            float2 v;
                            if (POST) v = D.lp_vel[aidx];
                            else v = W.lpc[tid];
                            orca_post(A, v.x, v.y, D.dt, D.inv_dt, d0, d1, &hint);
                            break;
                        }
                    }
                    a0 = (float)d0;
                    a1 = (float)d1;
                }
                asm volatile("" :: "v"(a0), "v"(a1));
                WAVETRACE(t, 15);
                if (TWO && !OBST && !POST) {
                    // M = 20 and the run-time-M kernels are held to 128 VGPRs (the whole launch co-resident) and the compiler kept the
                    // hint's three doubles alive through take_action in SCRATCH memory: two dependent round trips of ~600 cycles on the
                    // chain.  The LP scratch is dead here (every LP wave is done, phase A's keys come after the next barrier): the hint
                    // waits there instead (an LDS round trip; the empty statement keeps the compiler from forwarding the stores)
                    HeadingHint* hl = reinterpret_cast<HeadingHint*>(W.lp3) + tid;
                    *hl = hint;
                    asm volatile("" ::: "memory");
                    moved = take_action<false>(A, a0, a1, D.dt, hl);
                } else
                moved = take_action<false>(A, a0, a1, D.dt, &hint);
            }
            PMARK("D_s1_end");
            STAMP(3);
            WAVETRACE(t, 4);
        } else if (t > 0) {
            prio_rows3();
            PMARK("D_rows_begin");
            observation_chunks3<OBST, (ONE ? 8 : CAGYM_LASER_BATCH_ROLLOUT)>(D, W, o_prev, npairs, M, MP, K, C.wpw, C.worlds_valid, inv_m, ko, AS);
            prio_chain3(lagging);
            WAVETRACE(t, 4);
        }
        PMARK("D_end_barrierX");
        __syncthreads();  // rows of step t-1 are out: the moved state may replace the old one
        WGTRACE1(25);
        WAVETRACE(t, 5);
        if (s1_lane) {
            lds3_store_s1(W, A, tid);
            W.tmoved[tid] = moved ? 1 : 0;
            W.trvo[tid] = live_rvo(A.st, true);
        } else if (agent_lane) {
            W.tmoved[tid] = 0;
        }
        if (OVL && agent_lane) {  // the late arrays start their life here, in the bytes of the half-plane rows the linear programs are done with
            W.gmin[tid] = CAGYM_GAP_INF;
            W.hit[tid * MP + C.slot] = 0;
            W.keys[tid * MP + C.slot] = -INFINITY;
            for (int l = M; l < MP; l++) {
                W.hit[tid * MP + l] = 0;
                W.keys[tid * MP + l] = -INFINITY;
            }
        }
        if (tid == NT - 1) { W.flag[1] = 0; W.flag[2] = 0; W.flag[4] = W.flag[3]; W.flag[3] = 0; W.flag[7] = 0; W.flag[8] = 0; W.flag[9] = 0; W.flag[10] = 0; }
        __syncthreads();
        WGTRACE1(26);
        STAMP(4);
        WAVETRACE(t, 6);
        PMARK("A_begin");
        // ---- phase A: pair distances, collision tests, OAS sort keys, fp32 squared distances; the last wave first prepares
        //      the next step's LP inputs (Dynamics.update_ego_frame waits for phase B: nothing before the rows needs it) -------
        if (tid >= NT - CAGYM_WAVE) {
            const int a = tid - (NT - CAGYM_WAVE);
            if (a < nagents && any_rvo && t + 1 < n_steps) ego_lp_inputs3<OBST>(D, W, a, M, AS, ko, inv_m);
            WAVETRACE(t, 7);
        }
        if (OBST && D.map_bits && agent_lane) wall_prep3(D, W, tid, ko, inv_m, M);  // wave 0, beside the last wave's LP inputs
        if (dsq_shared && agent_lane) {  // the key rows share their bytes with the LP scratch: own slot and padding again
            W.dsq[tid * MP + C.slot] = make_uint2((uint32_t)C.slot, 0x7f800000u);
            for (int l = M; l < MP; l++) W.dsq[tid * MP + l] = make_uint2((uint32_t)l, 0x7f800000u);
        }
        PMARK("A_pairs_begin");
        for (int p = tid; p < nup; p += NT) pair_distances3<MT, true, !ONE>(D, W, p, M, MP);
        PMARK("A_pairs_end");
        WAVETRACE(t, 8);
        const bool obst_lines = OBST && ko > 0 && any_rvo && t + 1 < n_steps;
        if (OBST && (D.map_bits || obst_lines)) __syncthreads();
        WGTRACE1(40);
        if (OBST && D.map_bits) wall_rows3<NWAVES>(D, W, nagents, inv_m);
        WGTRACE1(41);
        if (obst_lines) obstacle_lines_phase3(D, W, M, AS, ko, nagents, inv_m);
        __syncthreads();
        WGTRACE1(27);
        STAMP(5);
        WAVETRACE(t, 9);
        PMARK("B_begin");
        // ---- phase B: wave 0 = S2 (_compute_rewards env.py:502-567, _check_which_agents_done :711-738, auto-reset);
        //      waves 1.. = ORCA half-planes of step t+1 (wave 0 takes the pairs beyond their lanes afterwards) ----------------
        const bool more = t + 1 < n_steps;
        const int full_pairs = nup - nup % (NT - CAGYM_WAVE), tail_pairs = nup - full_pairs;
        if (wave == 0) {
            if (agent_lane) {
                float reward = 0.f;
                Agent S;
                S.st = W.tst[tid];
                const double dmin = gap_of_key(W.gmin[tid]);  // the closest other agent (as the lower index of the pair, Q7), folded by phase A
                W.gmin[tid] = CAGYM_GAP_INF;                  // ... and consumed: the next step's pair lanes start from +inf
                if (C.valid && C.active) {
                    S.px = W.tpx[tid]; S.py = W.tpy[tid]; S.r = W.tr[tid];
                    bool coll_wall = false;
                    uint32_t hits = 0;
                    const uint32_t* hrow = reinterpret_cast<const uint32_t*>(W.hit + tid * MP);
                    for (int l4 = 0; l4 < MP; l4 += 4) hits |= hrow[l4 >> 2];
                    const bool coll_agent = hits != 0;
                    if (OBST && D.map_bits) coll_wall = W.lpk[tid] != 0;  // wall_prep3 / wall_rows3 in phase A
                    double r = -0.01;
                    if (S.st & CAGYM_FLAG_AT_GOAL) {
                        if (!(S.st & CAGYM_FLAG_WAS_AT_GOAL)) r = 3.0;
                    } else {
                        if (!(S.st & CAGYM_FLAG_WAS_IN_COLLISION)) {
                            if (coll_agent) { r = -10.0; S.st |= CAGYM_FLAG_IN_COLLISION; }
                            else if (coll_wall) { r = -0.25; S.st |= CAGYM_FLAG_IN_COLLISION; }
                            else if (dmin <= 0.2) r += -0.1 - dmin / 2.;
                        } else if (S.st & CAGYM_FLAG_RAN_OUT_OF_TIME) {
                            r += -10.0;
                        }
                    }
                    r = clipd(r, -10.0, 3.0) / (3.0 - (-10.0));
                    reward = (float)r;
                    if (S.st & (CAGYM_FLAG_AT_GOAL | CAGYM_FLAG_RAN_OUT_OF_TIME | CAGYM_FLAG_IN_COLLISION)) S.st |= CAGYM_FLAG_DONE;
                }
                const bool live = C.valid && C.active;
                const bool done = !live || (S.st & CAGYM_FLAG_DONE);
                const uint64_t wm = world_mask64(C);
                const uint64_t b_done = __ballot(done);
                const uint64_t b_learn = __ballot(done || ST_POLICY(S.st) != CAGYM_POL_LEARNING);
                bool go;
                if (D.go_mode == CAGYM_GO_ALL) go = (b_done & wm) == wm;
                else if (D.go_mode == CAGYM_GO_LEARNING) go = (b_learn & wm) == wm;
                else go = C.n > 0 ? ((b_done >> C.base) & 1ull) : true;
                if (C.valid) {
                    if (o.reward) o.reward[aidx] = reward;
                    if (o.flags) o.flags[aidx] = (uint8_t)(S.st & 0xffu);
                    if (C.slot == 0) {
                        if (o.game_over) o.game_over[C.world] = go ? 1 : 0;
                        ep_ret += reward;
                        ep_len += 1;
                    }
                }
                bool any_reset = false;
                if (AUTO_RESET) {
                    const bool rs = C.valid && go;
                    any_reset = __ballot(rs) != 0ull;
                    if (any_reset) {
                        float r0 = rs ? ep_ret : 0.f;
                        int l0 = rs ? ep_len : 0;
                        LaneCtx Cr = C;
                        Cr.valid = rs;
                        // (opaque: as a loop invariant the four 64-bit addresses of the world's statistics were computed in front of the step
                        // loop, found no register in the kernels held to 128 VGPRs - M = 20: 32 of its 36 bytes of scratch - and came back
                        // one dependent reload + load after the other inside this rare path)
                        asm volatile("" : "+v"(Cr.world));
                        fold_episode_stats(D, Cr, S, r0, l0);
                        if (rs) {
                            ep_ret = 0.f;
                            ep_len = 0;
                            C.episode += 1;
                            int sidx = (int)(((long long)C.world + (long long)C.episode * D.N) % D.S);
                            C.n = D.sc_nagents[sidx];
                            C.active = C.slot < C.n;
                            init_agent(D, S, sidx, C.slot, C.active);
                            lds3_store_agent(W, S, tid);
                            if (C.slot == 0) {
                                int wlq = C.wl;  // (opaque for the same reason: the hoisted LDS address was the last spilled value of the M = 20 kernel)
                                asm volatile("" : "+v"(wlq));
                                W.wn[wlq] = C.n;
                                W.wsc[wlq] = sidx;
                            }
                        }
                    }
                }
                W.tst[tid] = S.st;
                if (tid == 0) W.flag[0] = any_reset ? 1 : 0;
                // Dynamics.update_ego_frame (dynamics/Dynamics.py:14-28) of the agents S1 moved: only the observation rows
                // and the next S1 read it, and wave 0 is otherwise idle here until the half-plane lanes are done.  A world
                // that was just reset got its ego frame from init_agent.
                if (W.tmoved[tid] && !(AUTO_RESET && C.valid && go)) {
                    Agent E;
                    E.px = W.tpx[tid]; E.py = W.tpy[tid]; E.gx = W.tgx[tid]; E.gy = W.tgy[tid]; E.h = W.th[tid];
                    double prx, pry;
                    update_ego_frame(E, prx, pry);
                    W.tdg[tid] = E.dg; W.the[tid] = E.he; W.tprx[tid] = prx; W.tpry[tid] = pry;
                }
            }
            PMARK("B_s2_end");
            // a last partial round of at most one wave of pairs is wave 0's (it is done with S2 before the others finish)
            if (any_rvo && more && tail_pairs <= CAGYM_WAVE && tid < tail_pairs) half_planes3<MT>(D, W, full_pairs + tid, M, MP, AS, ko);
        } else if (any_rvo && more) {
            const int lim = tail_pairs <= CAGYM_WAVE ? full_pairs : nup;
            PMARK("B_hp_begin");
            for (int p = tid - CAGYM_WAVE; p < lim; p += NT - CAGYM_WAVE) half_planes3<MT>(D, W, p, M, MP, AS, ko);
        }
        if (OBST && ONE && o.laserscan) {
            // the one-step launch (every step of the VecEnv path; the roll-out kernels sit at their register limit and keep the scan
            // of their last step in the epilogue): no half-planes to build, waves 1.. would idle beside S2.  The
            // LaserScan only reads what S1 published (pose, radius, the world's rectangles), so it runs here - it was the longest part
            // of the epilogue (wave 0 joins behind S2); a world restarted by S2 is scanned again in the epilogue (W.flag[0]).
            laser_scan3<8>(D, W, o.laserscan, M, C.wpw, AS, inv_m, ko);
        }
        PMARK("B_end");
        WAVETRACE(t, 10);
        __syncthreads();
        WGTRACE1(28);
        STAMP(6);
        WAVETRACE(t, 11);
        PMARK("R_begin");
        // ---- rare: a world restarted on its next scenario -> everything derived from the old episode is rebuilt -------------
        if (AUTO_RESET && W.flag[0]) {
            if (OBST && tid == 0) { W.flag[7] = 0; W.flag[8] = 0; W.flag[9] = 0; W.flag[10] = 0; }  // the epilogue scans again (barriers below)
            if (OBST && ko > 0) {  // the restarted worlds' rectangles (also behind the last step: the epilogue's scan uses them)
                stage_rects3(D, W, C.wpw, C.worlds_valid);
                __syncthreads();
            }
            if (tid >= NT - CAGYM_WAVE) {
                const int a = tid - (NT - CAGYM_WAVE);
                if (a < nagents) {
                    const int wl = (int)__umulhi((uint32_t)a, inv_m);
                    W.trvo[a] = live_rvo(W.tst[a], (a - wl * M) < W.wn[wl]);
                    if (any_rvo && more) ego_lp_inputs3<OBST>(D, W, a, M, AS, ko, inv_m);
                }
            }
            for (int p = tid; p < nup; p += NT) pair_distances3<MT, false, !ONE>(D, W, p, M, MP);
            if (OBST && ko > 0 && any_rvo && more) {
                __syncthreads();
                obstacle_lines_phase3(D, W, M, AS, ko, nagents, inv_m);
            }
            __syncthreads();
            if (any_rvo && more)
                for (int p = tid; p < nup; p += NT) half_planes3<MT>(D, W, p, M, MP, AS, ko);
            __syncthreads();
        }
        PMARK("step_end");
        STAMP(7);
        WAVETRACE(t, 12);
        WGTRACE(2 + t);
    }
    // ---- epilogue: observation of the last step on every wave, agent records -> HBM ------------------------------------------
    {
        CagymOut o_last = out_slice3(out, n_steps - 1, (size_t)D.N, NM, M);
        if (OBST && ONE && !W.flag[0]) o_last.laserscan = nullptr;  // scanned beside S2 (phase B); again only after a restart
        observation_chunks3<OBST, (ONE ? 8 : CAGYM_LASER_BATCH_ROLLOUT)>(D, W, o_last, npairs, M, MP, K, C.wpw, C.worlds_valid, inv_m, ko, AS);
        if (C.valid) {
            const Agent A = lds3_load_agent(W, threadIdx.x);  // own lane's record, written by this lane
            store_agent(D, A, (size_t)C.world * M + C.slot, true);
            if (C.slot == 0) {
                D.ep_return[C.world] = ep_ret;
                D.ep_len[C.world] = ep_len;
                D.episode[C.world] = C.episode;
                D.n_agents[C.world] = C.n;
            }
        }
    }
}

// waves per SIMD the register budget is held to: 4 workgroups per CU for the specialisations whose LP groups hold one
// half-plane per lane (4096 worlds x 10 agents = 1024 workgroups = 4 per CU must be co-resident), 3 otherwise
#ifndef CAGYM_MINW_WIDE
#define CAGYM_MINW_WIDE 4  /* waves per SIMD held for the specialisations with two half-planes per LP lane (M = 20, generic) */
#endif
__host__ __device__ constexpr int cagym_min_waves3(int NT, int MT) { return NT > 256 ? 2 : ((MT > 0 && MT <= 10) ? 4 : CAGYM_MINW_WIDE); }

template <int NT, int MT, int WPWT, bool AUTO_RESET, bool OBST>
__global__ void __launch_bounds__(NT, OBST ? 2 : cagym_min_waves3(NT, MT)) k_rollout3(CagymDev D, int n_steps, CagymOut out, int any_rvo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WGTRACE(0);
    WGTRACE_VALUE(39, 0);
    // the OBST body is too large for the inliner's taste; called out of line it would get the device struct through scratch
    if (OBST) { [[clang::always_inline]] run_steps3<NT, MT, WPWT, AUTO_RESET, OBST>(D, smem, nullptr, out, n_steps, any_rvo != 0); }
    else run_steps3<NT, MT, WPWT, AUTO_RESET, OBST>(D, smem, nullptr, out, n_steps, any_rvo != 0);
    WGTRACE(38);
    WGTRACE_XCC();
}

// one step with external actions; the output buffers are NOT sliced (out_slice3 with t = 0 is the identity)
// (OBST: held to 128 VGPRs - with the time-shared LDS of the one-step launch (carve_lds3_ovl) four workgroups share a CU)
template <int NT, int MT, int WPWT, bool AUTO_RESET, bool OBST>
__global__ void __launch_bounds__(NT, OBST ? (NT > 256 ? 2 : 4) : cagym_min_waves3(NT, MT)) k_step3(CagymDev D, const float* ext, CagymOut out, int any_rvo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WGTRACE(0);
    WGTRACE_VALUE(39, 0);
    if (OBST) { [[clang::always_inline]] run_steps3<NT, MT, WPWT, AUTO_RESET, OBST, true>(D, smem, ext, out, 1, any_rvo != 0); }
    else run_steps3<NT, MT, WPWT, AUTO_RESET, OBST, true>(D, smem, ext, out, 1, any_rvo != 0);
    WGTRACE(38);
}
