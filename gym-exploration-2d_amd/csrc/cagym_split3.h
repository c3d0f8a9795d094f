// cagym_split3.h -- the split step: cagym_step_begin (k_step_pre3) + cagym_step_finish (k_step_post3).
//
// env.py:287-340 (_take_action) gathers the actions of ALL agents before any agent moves: an RVO ego's new velocity
// (policies/RVOPolicy.py:53-117) depends on the state BEFORE the step only, not on what the externally driven agents are about to
// do.  A caller whose external actions come from a device policy of its own (cfg4: cagym_ga3c_act, 74 us on the matrix cores)
// therefore does not have to run the two one after the other:
//
//   stream A:  cagym_step_begin   k_step_pre3   state -> LDS (the ten fields the ORCA half reads), obstacle + agent half-planes,
//                                               linear programs; 8 bytes per agent out (CagymDev::lp_vel)
//   stream B:  the caller's policy              (cagym_ga3c_act: selection, state vectors, fused forward) -> ext_actions
//   join:      cagym_step_finish  k_step_post3  run_steps3<.., ONE, POST>: S1 with every action in hand -> pair phase + wall test ->
//                                               S2 + LaserScan -> observations (-> auto-reset)
//
// Both halves are the fused one-step kernel's own phase functions in the same order on the same operands: a split step is
// bit-identical to cagym_step / cagym_step_autoreset (tests/test_split_step.py).  Each half carries only its own LDS: cfg4's
// 53.4 KB (three workgroups per CU, 2.67 rounds of its 2 048 workgroups) becomes 36 KB + 17 KB, four (and more) per CU.
#pragma once
#include "cagym_kernels3.h"

// ---- LDS of the PRE half -------------------------------------------------------------------------------------------------------
// head: 8 fp64 fields (px, py, vx, vy, r, gx, gy, pref), the two fp32 velocity pairs (lpv, lpc), six 4-byte columns (coop, trvo,
// nobl, lpr, lpk, busy), the per-world words and the flags
__host__ __device__ inline size_t cagym_pre3_head(int AS) { return (size_t)8 * AS * 8 + (size_t)2 * AS * 8 + (size_t)6 * AS * 4 + 96 * 4 + 16 * 4; }
// ONE scratch region serves three consumers that are alive one after the other: obstacle_lines_phase3's candidate lists (13 B per
// (ego, candidate)), then the neighbour keys `dsq` (pair lanes -> half-plane ranking), then the LP groups' projected lines
// (2 GW float4 per group: the obstacle lines of a linearProgram3 are read from their rows, only agent lines are projected)
__host__ __device__ inline size_t cagym_pre3_scratch(int M, int AS, int NT, int ko) {
    const size_t lists = a16((size_t)13 * ko * AS), keys = a16((size_t)AS * cagym_mp(M) * 8), proj = (size_t)2 * NT * 16;
    return lists > keys ? (lists > proj ? lists : proj) : (keys > proj ? keys : proj);
}
// the coverage bits of obstacle_lines_phase3 live in the agent rows of `sorted` (written by the half-plane lanes afterwards) when they fit
__host__ __device__ inline bool cagym_pre3_cov_aliased(int M, int AS, int ko) { return a16((size_t)ko * AS * 4) <= (size_t)(M - 1) * AS * 16; }
__host__ __device__ inline size_t cagym_lds3_pre_bytes(int M, int AS, int NT, int ko) {
    return a16(cagym_pre3_head(AS)) + (size_t)(ko + M - 1) * AS * 16 + cagym_pre3_scratch(M, AS, NT, ko) + (size_t)(AS / M) * (ko / 2) * 64 +
           (cagym_pre3_cov_aliased(M, AS, ko) ? 0 : a16((size_t)ko * AS * 4));
}
__device__ __forceinline__ Lds3 carve_lds3_pre(unsigned char* smem, int M, int AS, int NT, int ko) {
    Lds3 W = {};
    W.tpx = reinterpret_cast<double*>(smem);
    W.tpy = W.tpx + AS; W.tvx = W.tpy + AS; W.tvy = W.tvx + AS; W.tr = W.tvy + AS; W.tgx = W.tr + AS; W.tgy = W.tgx + AS; W.tpref = W.tgy + AS;
    W.lpv = reinterpret_cast<float2*>(W.tpref + AS);
    W.lpc = W.lpv + AS;
    W.tcoop = reinterpret_cast<float*>(W.lpc + AS);
    W.trvo = reinterpret_cast<int*>(W.tcoop + AS);
    W.nobl = W.trvo + AS;
    W.lpr = reinterpret_cast<float*>(W.nobl + AS);
    W.lpk = reinterpret_cast<int*>(W.lpr + AS);
    W.busy = W.lpk + AS;
    W.wn = W.busy + AS;
    W.wsc = W.wn + 32;
    W.wnob = W.wsc + 32;
    W.flag = W.wnob + 32;
    unsigned char* u = smem + a16(cagym_pre3_head(AS));
    W.sorted = reinterpret_cast<float4*>(u);
    u += (size_t)(ko + M - 1) * AS * 16;
    W.lp3 = reinterpret_cast<float4*>(u);
    W.dsq = reinterpret_cast<uint2*>(u);
    u += cagym_pre3_scratch(M, AS, NT, ko);
    W.rect = reinterpret_cast<float4*>(u);
    u += (size_t)(AS / M) * (ko / 2) * 64;
    W.cov = cagym_pre3_cov_aliased(M, AS, ko) ? reinterpret_cast<uint32_t*>(W.sorted + (size_t)ko * AS) : reinterpret_cast<uint32_t*>(u);
    return W;
}

// ---- PRE half: the ORCA solve of every live RVO ego of the workgroup's worlds ------------------------------------------------------
template <int NT, int MT, int WPWT, bool OBST>
__device__ inline void run_rvo_pre3(const CagymDev& D, unsigned char* smem) {
    constexpr int NWAVES = NT / CAGYM_WAVE;
    constexpr int GW = cagym_gw3(MT), NG = NT / GW;
    constexpr bool TWO = cagym_two3(MT);
    const int M = MT ? MT : D.M, MP = cagym_mp(M);
    const int AS = cagym_as(M, WPWT);
    const int ko = OBST ? D.ko : 0;
    const Lds3 W = carve_lds3_pre(smem, M, AS, NT, ko);
    const LaneCtx C = make_ctx2(D, M, WPWT ? WPWT : CAGYM_WAVE / M);
    const uint32_t inv_m = (uint32_t)(0x100000000ull / (uint32_t)M) + 1u;
    const int nagents = C.wpw * M, nup = C.wpw * (M * (M - 1) / 2);
    const int tid = threadIdx.x;
    const bool agent_lane = tid < nagents;
    const bool mine = agent_lane && C.valid;
    const size_t aidx = (size_t)(mine ? C.world : 0) * M + (mine ? C.slot : 0);
    WGTRACE(0);
    if (agent_lane) {
        double px = 0, py = 0, vx = 0, vy = 0, r = 0, gx = 0, gy = 0, pref = 0, coop = 0;
        uint32_t st = 0;
        if (mine) {
            px = D.px[aidx]; py = D.py[aidx]; vx = D.vx[aidx]; vy = D.vy[aidx]; r = D.radius[aidx];
            gx = D.gx[aidx]; gy = D.gy[aidx]; pref = D.pref[aidx]; coop = D.coop[aidx];
            st = D.status[aidx];
        }
        W.tpx[tid] = px; W.tpy[tid] = py; W.tvx[tid] = vx; W.tvy[tid] = vy; W.tr[tid] = r;
        W.tgx[tid] = gx; W.tgy[tid] = gy; W.tpref[tid] = pref;
        W.tcoop[tid] = (float)coop;
        W.trvo[tid] = live_rvo(st, mine && C.active);
        W.nobl[tid] = 0;
        if (C.wl < C.wpw && C.slot == 0) {
            W.wn[C.wl] = C.valid ? C.n : 0;
            W.wsc[C.wl] = C.valid ? (int)(((long long)C.world + (long long)C.episode * D.N) % D.S) : 0;
        }
    }
    if (tid < 16) W.flag[tid] = 0;
    __syncthreads();
    WGTRACE(20);
    if (OBST && ko > 0) {
        stage_rects3(D, W, C.wpw, C.worlds_valid);
        __syncthreads();
    }
    WGTRACE(21);
    if (agent_lane) ego_lp_inputs3<OBST>(D, W, tid, M, AS, ko, inv_m);
    if (OBST && ko > 0) {
        __syncthreads();
        obstacle_lines_phase3(D, W, M, AS, ko, nagents, inv_m);
    }
    WGTRACE(22);
    // the neighbour keys take the bytes of the candidate lists: those are dead once every lane has passed the last barrier inside
    // obstacle_lines_phase3 (its last sub-step, on the last wave, reads the coverage bits and the rows only)
    if (agent_lane) {
        W.dsq[tid * MP + C.slot] = make_uint2((uint32_t)C.slot, 0x7f800000u);
        for (int l = M; l < MP; l++) W.dsq[tid * MP + l] = make_uint2((uint32_t)l, 0x7f800000u);
    }
    for (int p = tid; p < nup; p += NT) pair_dsq3<MT>(W, p, M, MP);
    __syncthreads();
    WGTRACE(23);
    for (int p = tid; p < nup; p += NT) half_planes3<MT>(D, W, p, M, MP, AS, ko);
    __syncthreads();
    WGTRACE(24);
    // linearProgram2/3 of every busy ego on a GW-lane group (phase C of run_steps3; every wave builds the same list)
    {
        const int lane = tid & (CAGYM_WAVE - 1);
        const bool fl = lane < nagents && W.busy[lane] != 0;
        const unsigned long long bm = __ballot(fl);
        const int cnt = __popcll(bm);
        if (fl) W.lpk[__popcll(bm & ((1ull << lane) - 1ull))] = lane;
        WGTRACE_BUSY(cnt);
        const int g = tid / GW, j = tid & (GW - 1);
        for (int base = 0; base < cnt; base += NG) {
            const int idx = base + g;
            if (idx < cnt) {
                const int a = W.lpk[idx];
                const int wl = (int)__umulhi((uint32_t)a, inv_m);
                const int n = W.wn[wl];
                const int nn = (n - 1) < D.maxnb ? (n - 1) : D.maxnb;
                const float2 pv = W.lpv[a];
                const float rad = W.lpr[a];
                float vx, vy;
                float4* P = W.lp3 + 2 * (tid & ~(GW - 1));
                if (OBST) {
                    const int nol = W.nobl[a];
                    if (__ballot(nol + nn > 2 * GW) != 0ull) orca_lp_group_n<GW, 4, true>(W.sorted, P, a, j, nol, nn, ko, rad, pv.x, pv.y, vx, vy, AS, nullptr);
                    else orca_lp_group_n<GW, 2, true>(W.sorted, P, a, j, nol, nn, ko, rad, pv.x, pv.y, vx, vy, AS, nullptr);
                } else {
                    orca_lp_group<GW, TWO, (MT > 0 ? MT - 1 : 0)>(W.sorted, P, a, j, nn, rad, pv.x, pv.y, vx, vy, AS, nullptr);
                }
                if (j == 0) W.lpc[a] = make_float2(vx, vy);
            }
        }
    }
    __syncthreads();
    WGTRACE(25);
    if (mine) D.lp_vel[aidx] = W.lpc[tid];
    (void)NWAVES;
}

template <int NT, int MT, int WPWT, bool OBST>
__global__ void __launch_bounds__(NT, 4) k_step_pre3(CagymDev D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    run_rvo_pre3<NT, MT, WPWT, OBST>(D, smem);
    WGTRACE(38);
}

// the second half: one step with the RVO velocities of k_step_pre3 and the caller's external actions (run_steps3, POST)
template <int NT, int MT, int WPWT, bool AUTO_RESET, bool OBST>
__global__ void __launch_bounds__(NT, 4) k_step_post3(CagymDev D, const float* ext, CagymOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WGTRACE(0);
    WGTRACE_VALUE(39, 0);
    [[clang::always_inline]] run_steps3<NT, MT, WPWT, AUTO_RESET, OBST, true, true>(D, smem, ext, out, 1, false);
    WGTRACE(38);
}
