// cagym_kernels2.h -- phase-split env.step() kernels (generation 2).
//
// Why: with one lane per agent a 4096 x 10 batch is 683 lone waves on 1024 SIMDs; SQ counters showed
// ~11.5 k instructions per wave-step, ~85 % of them the O(M^2) neighbour loops, 47 % of cycles in waits
// (profiles/r1/history/*pmc_sq.txt).  Here a workgroup of NT threads owns a few whole worlds, the agent records
// live in LDS, and a step alternates between
//   S phases: per-AGENT work on wave 0 (action maps + dynamics; reward / done / reset),
//   P phases: per-PAIR work on all NT lanes (ORCA half-planes per unordered pair; pair distances and collision
//             tests per unordered pair; OAS ranks and rows per directed pair, stored straight to HBM), and
//   the LP phase: linearProgram2/3 of every ego with a violated half-plane on an 8-lane group (lane j <-> line j,
//             DPP reductions),
// separated by workgroup barriers; idle lanes of the pair phases run the ego-frame update and prepare the next
// step's LP inputs.  DESIGN.md section 4 has the full list and the measurements behind each choice.  Arithmetic is
// shared with generation 1 (cagym_device.h, cagym_orca.h): both generations produce bit-identical results
// (tests/test_hip_parity.py).
#pragma once
#include "cagym_kernels.h"

// Diagnostic build only (-DCAGYM_STAMPS, never the shipped library): thread 0 of workgroup 0 accumulates
// s_memtime deltas per phase into g_stamps; read back with cagym_debug_stamps().  The stamp values leave
// the kernel only through this buffer and feed no output.
#ifdef CAGYM_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(i)                                                                  \
    do {                                                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                \
            unsigned long long _t = __builtin_amdgcn_s_memtime();                 \
            g_stamps[i] += _t - stamp_prev;                                       \
            stamp_prev = _t;                                                      \
        }                                                                         \
    } while (0)
#define STAMP_BEGIN() unsigned long long stamp_prev = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i) do { } while (0)
#define STAMP_BEGIN() do { } while (0)
#endif

// Second diagnostic build (-DCAGYM_WGTRACE, tools/launch_cost.py): thread 0 of EVERY workgroup records the 100 MHz
// s_memrealtime clock at kernel entry, after the prologue, after each of the first 36 steps and at exit, plus its XCC id,
// into g_wgtrace (read back with cagym_debug_wgtrace()).  Same rule: the values feed no output.
#ifdef CAGYM_WGTRACE
#define CAGYM_WGTRACE_MAXWG 4096
#define CAGYM_WGTRACE_W 40
__device__ unsigned long long g_wgtrace[CAGYM_WGTRACE_MAXWG * CAGYM_WGTRACE_W];
#define WGTRACE(slot)                                                                                       \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG && (slot) < CAGYM_WGTRACE_W)               \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + (slot)] = __builtin_amdgcn_s_memrealtime();            \
    } while (0)
#define WGTRACE_BUSY(cnt)                                                                                   \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG)                                           \
            g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + 39] += (unsigned long long)(cnt) << 8;                 \
    } while (0)
#else
#define WGTRACE(slot) do { } while (0)
#define WGTRACE_BUSY(cnt) do { } while (0)
#endif

#ifndef CAGYM_GW10
#define CAGYM_GW10 8  // lanes per ORCA LP group when M <= 10 (nn <= 9 half-planes)
#endif

struct Lds2 {
    // [64] each: the agent record lives HERE between phases (registers only inside the S phases)
    double *tpx, *tpy, *tvx, *tvy, *tr, *tprx, *tpry;
    double *th, *the, *tdg, *ttrem, *tt, *tgx, *tgy, *tpref, *tspeed, *tdh, *taux0, *taux1, *tcoopd;
    float2* tact;
    float* tcoop;
    uint32_t* tst;
    int* tstep;
    int* tmoved;                                         // [64] the agent moved in this step's S1 (ego frame still to be updated)
    int* wn;                                             // [32] agents per world of this workgroup
    int* flag;                                           // [4]  0: any world reset this step
    float2* lpv;                                         // [64] preferred (optimisation) velocity of each ego
    int* lpk;                                            // [64] compact list of the busy egos (LP phase)
    float* lpr;                                          // [64] maxSpeed of the ego (LP radius)
    unsigned long long* lpmask;                          // [2]  padding (keeps the 8-byte alignment of what follows)
    float2* lpc;                                         // [64] pref velocity clipped to maxSpeed = LP start; LP result afterwards
    int* busy;                                           // [64] some half-plane of the ego is violated by its LP start
    // union, pre-move view.  Rows are padded to MP = roundup(M, 4) entries and unused entries hold
    // sentinels, so the rank loops are branch-free 16-byte LDS reads.
    float* dsq;      // [64*MP]   squared centre distance ego->neighbour (+inf = not a neighbour)
    float4* lines;   // [64*MP]   half-plane of (ego, neighbour), unsorted
    float4* sorted;  // [10][64]  half-planes nearest-first
    // union, post-move view
    double* keys;  // [64*MP]   OAS sort key (-inf = not observed)
    double* gap;   // [64*MP]   d - (r_i + r_j) for the lower index of a pair, else +inf
    uint8_t* hit;  // [64*MP]   pair collides
    float* oas;    // [64*(M-1)*10]
};

__host__ __device__ inline size_t a16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline int cagym_mp(int M) { return (M + 3) & ~3; }

// AS = agent slots per workgroup (worlds per workgroup x M, rounded up to 4); 64 when a full wave is used
__host__ __device__ inline int cagym_as(int M, int wpw) { return wpw > 0 ? ((wpw * M + 3) & ~3) : 64; }

__host__ __device__ inline size_t cagym_lds2_bytes(int M, int AS = 64) {
    const size_t MP = cagym_mp(M);
    size_t head = 20 * AS * 8 + AS * 8 + AS * 4 + AS * 4 + AS * 4 + AS * 4 + 32 * 4 + 16 + AS * 8 + AS * 4 + AS * 4 + 16 + AS * 8 + AS * 4;
    // pre-move view: dsq, lines (an ego's row doubles as its linearProgram3 scratch once its group has sorted it),
    // sorted.  post-move view: keys, gap, hit (the OAS rows go straight to HBM).
    size_t pre = AS * MP * 4 + AS * MP * 16 + (size_t)CAGYM_MAXNB * AS * 16;
    size_t post = 2 * AS * MP * 8 + AS * MP;
    return a16(head) + (pre > post ? pre : post);
}

__device__ __forceinline__ Lds2 carve_lds2(unsigned char* smem, int M, int AS = 64) {
    Lds2 W;
    const size_t MP = cagym_mp(M);
    W.tpx = reinterpret_cast<double*>(smem);
    W.tpy = W.tpx + AS; W.tvx = W.tpy + AS; W.tvy = W.tvx + AS; W.tr = W.tvy + AS; W.tprx = W.tr + AS; W.tpry = W.tprx + AS;
    W.th = W.tpry + AS; W.the = W.th + AS; W.tdg = W.the + AS; W.ttrem = W.tdg + AS; W.tt = W.ttrem + AS;
    W.tgx = W.tt + AS; W.tgy = W.tgx + AS; W.tpref = W.tgy + AS; W.tspeed = W.tpref + AS; W.tdh = W.tspeed + AS;
    W.taux0 = W.tdh + AS; W.taux1 = W.taux0 + AS; W.tcoopd = W.taux1 + AS;
    W.tact = reinterpret_cast<float2*>(W.tcoopd + AS);
    W.tcoop = reinterpret_cast<float*>(W.tact + AS);
    W.tst = reinterpret_cast<uint32_t*>(W.tcoop + AS);
    W.tstep = reinterpret_cast<int*>(W.tst + AS);
    W.tmoved = W.tstep + AS;
    W.wn = W.tmoved + AS;
    W.flag = W.wn + 32;
    W.lpv = reinterpret_cast<float2*>(W.flag + 4);
    W.lpk = reinterpret_cast<int*>(W.lpv + AS);
    W.lpr = reinterpret_cast<float*>(W.lpk + AS);
    W.lpmask = reinterpret_cast<unsigned long long*>(W.lpr + AS);
    W.lpc = reinterpret_cast<float2*>(W.lpmask + 2);
    W.busy = reinterpret_cast<int*>(W.lpc + AS);
    size_t head = 20 * AS * 8 + AS * 8 + AS * 4 + AS * 4 + AS * 4 + AS * 4 + 32 * 4 + 16 + AS * 8 + AS * 4 + AS * 4 + 16 + AS * 8 + AS * 4;
    unsigned char* u = smem + a16(head);
    W.dsq = reinterpret_cast<float*>(u);
    W.lines = reinterpret_cast<float4*>(u + AS * MP * 4);
    W.sorted = W.lines + AS * MP;
    W.keys = reinterpret_cast<double*>(u);
    W.gap = W.keys + AS * MP;
    W.hit = reinterpret_cast<uint8_t*>(W.gap + AS * MP);
    return W;
}

// fields that change when an agent moves
__device__ __forceinline__ void lds_store_moved(const Lds2& W, const Agent& A, int lane) {
    W.tpx[lane] = A.px; W.tpy[lane] = A.py; W.tvx[lane] = A.vx; W.tvy[lane] = A.vy;
    W.tprx[lane] = A.prx; W.tpry[lane] = A.pry;
    W.th[lane] = A.h; W.the[lane] = A.he; W.tdg[lane] = A.dg; W.ttrem[lane] = A.trem; W.tt[lane] = A.t;
    W.tspeed[lane] = A.speed; W.tdh[lane] = A.dh; W.taux0[lane] = A.aux0; W.taux1[lane] = A.aux1;
    W.tact[lane] = make_float2(A.a0, A.a1);
    W.tst[lane] = A.st;
    W.tstep[lane] = A.step;
}
__device__ __forceinline__ void lds_store_agent(const Lds2& W, const Agent& A, int lane) {
    lds_store_moved(W, A, lane);
    W.tr[lane] = A.r; W.tgx[lane] = A.gx; W.tgy[lane] = A.gy; W.tpref[lane] = A.pref;
    W.tcoopd[lane] = A.coop;
    W.tcoop[lane] = (float)A.coop;
}
__device__ __forceinline__ Agent lds_load_agent(const Lds2& W, int lane) {
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

// prefVelocity = pref_speed (goal - pos) / |goal - pos| and maxSpeed of agent a (RVOPolicy.py:65-85, as orca_ego),
// from the agent record in LDS into the slots the next LP phase reads (lpv doubles as the LP result afterwards).
__device__ __forceinline__ void publish_pref_velocity(const Lds2& W, int a) {
    const double gx = W.tgx[a] - W.tpx[a], gy = W.tgy[a] - W.tpy[a];
    const double pref = W.tpref[a];
    const double sc = pref / norm2(gx, gy);
    const float ox = (float)(sc * gx), oy = (float)(sc * gy), radius = (float)pref;
    W.lpv[a] = make_float2(ox, oy);
    W.lpr[a] = radius;
    // linearProgram2's starting point: the optimisation velocity clipped to maxSpeed.  An ego none of whose half-planes
    // is violated there keeps it as its new velocity (the LP never moves it), so only the others ("busy") get an LP group.
    float cx = ox, cy = oy;
    if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);
        cx = ox * inv * radius;
        cy = oy * inv * radius;
    }
    W.lpc[a] = make_float2(cx, cy);
    W.busy[a] = 0;
}

// One env.step() of the workgroup's worlds.  Agent registers A live on wave 0 (tid < 64) only; the pair
// phases keep nothing in registers across barriers (everything is re-read from LDS).
template <int NT, int MT, int WPWT, bool AUTO_RESET>
__device__ inline void step_core2(const CagymDev& D, const Lds2& W, LaneCtx& C, const float* ext, const CagymOut& out,
                                  float& ep_ret, int& ep_len, bool any_rvo) {
    int tid = threadIdx.x;
    // opaque per step: keeps the compiler from hoisting every (agent, neighbour) index derived from tid out of
    // the rollout's step loop (that hoisting cost ~100 VGPRs and spilled; recomputing is a few ALU ops)
    asm volatile("" : "+v"(tid));
    const int M = MT ? MT : D.M, K = M - 1, MP = cagym_mp(M);  // MT > 0: M is a compile-time constant
    const int AS = cagym_as(M, WPWT);                           // LDS agent stride (64, or worlds-per-WG x M)
    const uint32_t inv_m = (uint32_t)(0x100000000ull / (uint32_t)M) + 1u;
    const bool agent_lane = tid < C.wpw * M;  // wave 0, one lane per agent slot of this workgroup
    const size_t aidx = (size_t)C.world * M + C.slot;
    const int npairs = C.wpw * M * M;
    STAMP_BEGIN();
    // ---- S0: the previous step's staging reads are done before the union is rewritten ------------------
    __syncthreads();
    STAMP(0);
    // ---- P1: ORCA half-planes, one lane per (ego, neighbour) -------------------------------------------
    if (any_rvo) {
        if (MT > 0) {
            // compile-time M: one lane per UNORDERED pair writes both half-planes (orca_pair is exactly odd
            // under ego <-> other), halving the lanes and rounds of this phase
            if (agent_lane) {
                W.dsq[tid * MP + C.slot] = INFINITY;
                for (int l = M; l < MP; l++) W.dsq[tid * MP + l] = INFINITY;
            }
            const int nup = C.wpw * UnorderedPairs<MT>::N;
            for (int p = tid; p < nup; p += NT) {
                const UPair q = UnorderedPairs<MT>::of(p);
                const int n = W.wn[q.wl];
                const int a = q.wl * M + q.i, b = q.wl * M + q.j;
                const uint32_t sa = W.tst[a], sb = W.tst[b];
                const bool both = q.i < n && q.j < n;
                const bool on_a = both && ST_POLICY(sa) == CAGYM_POL_RVO && !(sa & CAGYM_FLAG_DONE);
                const bool on_b = both && ST_POLICY(sb) == CAGYM_POL_RVO && !(sb & CAGYM_FLAG_DONE);
                float dq = INFINITY;
                if (on_a || on_b) {
                    const float vax = (float)W.tvx[a], vay = (float)W.tvy[a];
                    const OrcaPair g = orca_pair((float)W.tpx[a], (float)W.tpy[a], vax, vay,
                                                 (float)((1 + 15e-2) * W.tr[a]), (float)D.dt, W.tpx[b], W.tpy[b],
                                                 W.tvx[b], W.tvy[b], W.tr[b]);
                    dq = g.d2;
                    if (on_a) {
                        const float c = W.tcoop[a];
                        const float4 ln = make_float4(vax + c * g.ux, vay + c * g.uy, g.zx, g.zy);
                        W.lines[a * MP + q.j] = ln;
                        const float2 s0 = W.lpc[a];
                        if (detf(ln.z, ln.w, ln.x - s0.x, ln.y - s0.y) > 0.0f) W.busy[a] = 1;
                    }
                    if (on_b) {
                        const float c = W.tcoop[b];
                        const float4 ln = make_float4((float)W.tvx[b] - c * g.ux, (float)W.tvy[b] - c * g.uy, -g.zx, -g.zy);
                        W.lines[b * MP + q.i] = ln;
                        const float2 s0 = W.lpc[b];
                        if (detf(ln.z, ln.w, ln.x - s0.x, ln.y - s0.y) > 0.0f) W.busy[b] = 1;
                    }
                }
                W.dsq[a * MP + q.j] = on_a ? dq : INFINITY;
                W.dsq[b * MP + q.i] = on_b ? dq : INFINITY;
            }
        } else
        for (int p = tid; p < npairs; p += NT) {
            const PairIdx q = pair_of(p, M, inv_m);
            const int n = W.wn[q.wl];
            const uint32_t st = W.tst[q.a];
            const bool on = q.sl < n && q.j < n && q.j != q.sl && ST_POLICY(st) == CAGYM_POL_RVO && !(st & CAGYM_FLAG_DONE);
            float dq = INFINITY;
            if (on) {
                const int b = q.a - q.sl + q.j;
                const float pex = (float)W.tpx[q.a], pey = (float)W.tpy[q.a];
                const float dx = pex - (float)W.tpx[b], dy = pey - (float)W.tpy[b];
                dq = dx * dx + dy * dy;
                const float4 ln = orca_line(pex, pey, (float)W.tvx[q.a], (float)W.tvy[q.a],
                                            (float)((1 + 15e-2) * W.tr[q.a]), W.tcoop[q.a], (float)D.dt,
                                            W.tpx[b], W.tpy[b], W.tvx[b], W.tvy[b], W.tr[b]);
                W.lines[q.a * MP + q.j] = ln;
                const float2 s0 = W.lpc[q.a];
                if (detf(ln.z, ln.w, ln.x - s0.x, ln.y - s0.y) > 0.0f) W.busy[q.a] = 1;
            }
            W.dsq[q.a * MP + q.j] = dq;
            if (q.j == M - 1)
                for (int l = M; l < MP; l++) W.dsq[q.a * MP + l] = INFINITY;
        }
        __syncthreads();
        STAMP(1);
    }
    // ---- S1: _take_action (env.py:287-340).  RVO: linearProgram2 (+ linearProgram3 when infeasible) of every
    //      live RVO agent on a GW-lane group, lane j <-> half-plane j; the agents come from the compact list the
    //      previous S2 (or the kernel prologue) published ----------------------------------------------------
    if (any_rvo) {
        constexpr int GW = MT > 0 && MT <= 5 ? 4 : (MT > 0 && MT <= 10 ? CAGYM_GW10 : 16), NG = NT / GW;
        // every wave builds the same compact list of busy egos (identical values to identical LDS slots)
        int cnt;
        {
            const int lane = tid & (CAGYM_WAVE - 1);
            const bool fl = lane < C.wpw * M && W.busy[lane] != 0;
            const unsigned long long bm = __ballot(fl);
            cnt = __popcll(bm);
            if (fl) W.lpk[__popcll(bm & ((1ull << lane) - 1ull))] = lane;
        }
        WGTRACE_BUSY(cnt);
        const int g = tid / GW, j = tid & (GW - 1);
        // busy egos are packed into as few waves as possible (dealing them round-robin over the waves was slower at
        // every launch size: 301 vs 312 M env-steps/s at 4096 worlds, 507 vs 521 M at 65536)
        for (int base = 0; base < cnt; base += NG) {
            const int idx = base + g;
            if (idx < cnt) {
                const int a = W.lpk[idx];
                const int wl = (int)__umulhi((uint32_t)a, inv_m);
                const int n = W.wn[wl];
                const int nn = (n - 1) < CAGYM_MAXNB ? (n - 1) : CAGYM_MAXNB;
                // the group ranks its ego's neighbours (nearest first, ties by lower index: Agent::insertAgentNeighbor)
                // and scatters the half-planes into sorted order; producer and consumer are the same wave, whose LDS
                // operations execute in order, so no barrier separates the scatter from the reads below
                {
                    const float4* row = reinterpret_cast<const float4*>(W.dsq + a * MP);
                    for (int sl = j; sl < M; sl += GW) {
                        const float dq = W.dsq[a * MP + sl];
                        if (!(dq < INFINITY)) continue;
                        int rank = 0;
                        for (int l4 = 0; l4 < MP; l4 += 4) {
                            const float4 v = row[l4 >> 2];
                            rank += (v.x < dq) || (v.x == dq && l4 + 0 < sl);
                            rank += (v.y < dq) || (v.y == dq && l4 + 1 < sl);
                            rank += (v.z < dq) || (v.z == dq && l4 + 2 < sl);
                            rank += (v.w < dq) || (v.w == dq && l4 + 3 < sl);
                        }
                        if (rank < CAGYM_MAXNB) W.sorted[rank * AS + a] = W.lines[a * MP + sl];
                    }
                }
                // prefVelocity and maxSpeed of the ego were published by publish_pref_velocity (idle lanes of the
                // previous step's last phase, or the kernel prologue)
                const float2 pv = W.lpv[a];
                const float rad = W.lpr[a];
                float vx, vy;
                STAMP(10);  // list read + ranking
                // linearProgram3 scratch: the ego's own (now dead) row of unsorted half-planes, MP >= nn entries
                orca_lp_group<GW>(W.sorted, W.lines + a * MP, a, j, nn, rad, pv.x, pv.y, vx, vy, AS);
                if (j == 0) W.lpc[a] = make_float2(vx, vy);
                STAMP(11);  // LP of group 0's agent
            }
        }
        __syncthreads();
        STAMP(8);
    }
    if (agent_lane && C.valid && C.active) {
        Agent A = lds_load_agent(W, tid);
        float a0 = 0.f, a1 = 0.f;
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
                    const float2 v = W.lpc[tid];  // LP result, or the clipped preferred velocity of an ego that needed none
                    orca_post(A, v.x, v.y, D.dt, d0, d1);
                    break;
                }
            }
            a0 = (float)d0;
            a1 = (float)d1;
        }
        const bool moved = take_action<false>(A, a0, a1, D.dt);
        lds_store_moved(W, A, tid);  // nobody reads the tile here (P1 / LP readers are behind their barriers)
        W.tmoved[tid] = moved ? 1 : 0;
    } else if (agent_lane) {
        W.tmoved[tid] = 0;
    }
    __syncthreads();  // post-move tile visible; LP scratch (union, pre-move view) is dead
    STAMP(3);
    // ---- P2: pair distances, collision tests (env.py:630-655), OAS sort keys; the last wave (idle or lightly
    //      loaded in the unordered-pair loop) runs Dynamics.update_ego_frame of the agents that moved ---------------
    if (tid >= NT - CAGYM_WAVE) {
        const int a = tid - (NT - CAGYM_WAVE);
        if (a < C.wpw * M && W.tmoved[a]) {
            Agent E;
            E.px = W.tpx[a]; E.py = W.tpy[a]; E.gx = W.tgx[a]; E.gy = W.tgy[a]; E.h = W.th[a];
            double prx, pry;
            update_ego_frame(E, prx, pry);
            W.tdg[a] = E.dg; W.the[a] = E.he; W.tprx[a] = prx; W.tpry[a] = pry;
        }
    }
    if (MT > 0) {
        // one lane per unordered pair: the distance (fp64 sqrt) is shared by both directions
        if (agent_lane) {
            W.hit[tid * MP + C.slot] = 0;
            W.gap[tid * MP + C.slot] = INFINITY;
            W.keys[tid * MP + C.slot] = -INFINITY;
            for (int l = M; l < MP; l++) {
                W.hit[tid * MP + l] = 0;
                W.gap[tid * MP + l] = INFINITY;
                W.keys[tid * MP + l] = -INFINITY;
            }
        }
        const int nup = C.wpw * UnorderedPairs<MT>::N;
        for (int p = tid; p < nup; p += NT) {
            const UPair q = UnorderedPairs<MT>::of(p);
            const int n = W.wn[q.wl];
            const int lo = q.wl * M + (q.i < q.j ? q.i : q.j), hi = q.wl * M + (q.i < q.j ? q.j : q.i);
            const int slo = lo - q.wl * M, shi = hi - q.wl * M;
            double klo = -INFINITY, khi = -INFINITY, gp = INFINITY;
            uint8_t ht = 0;
            if (shi < n) {  // slo < shi < n
                const double dx = W.tpx[hi] - W.tpx[lo], dy = W.tpy[hi] - W.tpy[lo];
                const double d = norm2(dx, dy);
                const double rl = W.tr[lo], rh = W.tr[hi];
                const bool skip = ST_POLICY(W.tst[hi]) == CAGYM_POL_STATIC && !D.collide_static;  // env.py:643 (Q8)
                const double cr = rl + rh;
                ht = (!skip && d <= cr) ? 1 : 0;
                if (!skip) gp = d - cr;  // lower index only (Q7)
                klo = d - rl - rh;
                khi = d - rh - rl;
            }
            W.hit[lo * MP + shi] = ht;
            W.hit[hi * MP + slo] = ht;
            W.gap[lo * MP + shi] = gp;
            W.gap[hi * MP + slo] = INFINITY;
            W.keys[lo * MP + shi] = klo;
            W.keys[hi * MP + slo] = khi;
        }
    } else
    for (int p = tid; p < npairs; p += NT) {
        const PairIdx q = pair_of(p, M, inv_m);
        const int n = W.wn[q.wl];
        const bool on = q.sl < n && q.j < n && q.j != q.sl;
        double key = -INFINITY, gp = INFINITY;
        uint8_t ht = 0;
        if (on) {
            const int b = q.a - q.sl + q.j;
            const double dx = W.tpx[b] - W.tpx[q.a], dy = W.tpy[b] - W.tpy[q.a];
            const double d = norm2(dx, dy);
            const double ra = W.tr[q.a], rb = W.tr[b];
            const bool static_hi = ST_POLICY(q.j > q.sl ? W.tst[b] : W.tst[q.a]) == CAGYM_POL_STATIC;
            const bool skip = static_hi && !D.collide_static;  // env.py:643 (Q8)
            const double cr = (q.j > q.sl) ? (ra + rb) : (rb + ra);
            ht = (!skip && d <= cr) ? 1 : 0;
            if (!skip && q.j > q.sl) gp = d - cr;  // lower index only (Q7)
            key = d - ra - rb;
        }
        W.hit[q.a * MP + q.j] = ht;
        W.gap[q.a * MP + q.j] = gp;
        W.keys[q.a * MP + q.j] = key;
        if (q.j == M - 1)
            for (int l = M; l < MP; l++) {
                W.hit[q.a * MP + l] = 0;
                W.gap[q.a * MP + l] = INFINITY;
                W.keys[q.a * MP + l] = -INFINITY;
            }
    }
    __syncthreads();
    STAMP(4);
    // ---- S2: _compute_rewards (env.py:502-567), _check_which_agents_done (:711-738), auto-reset -----------
    if (agent_lane) {
        float reward = 0.f;
        Agent A;
        A.st = W.tst[tid];
        if (C.valid && C.active) {
            A.px = W.tpx[tid]; A.py = W.tpy[tid]; A.r = W.tr[tid];
            bool coll_wall = false;
            double dmin = INFINITY;
            uint32_t hits = 0;
            const uint32_t* hrow = reinterpret_cast<const uint32_t*>(W.hit + tid * MP);
            const double2* grow = reinterpret_cast<const double2*>(W.gap + tid * MP);
            for (int l4 = 0; l4 < MP; l4 += 4) {
                hits |= hrow[l4 >> 2];
                const double2 g0 = grow[l4 >> 1], g1 = grow[(l4 >> 1) + 1];
                dmin = fmin(dmin, fmin(fmin(g0.x, g0.y), fmin(g1.x, g1.y)));
            }
            const bool coll_agent = hits != 0;
            if (D.map_bits) {
                int sidx = (int)(((long long)C.world + (long long)C.episode * D.N) % D.S);
                if (D.sc_nobst[sidx] > 0)
                    coll_wall = wall_collision(D.map_bits + (size_t)sidx * CAGYM_MAPD * CAGYM_MAPW, A.px, A.py, A.r);
            }
            double r = -0.01;
            if (A.st & CAGYM_FLAG_AT_GOAL) {
                if (!(A.st & CAGYM_FLAG_WAS_AT_GOAL)) r = 3.0;
            } else {
                if (!(A.st & CAGYM_FLAG_WAS_IN_COLLISION)) {
                    if (coll_agent) { r = -10.0; A.st |= CAGYM_FLAG_IN_COLLISION; }
                    else if (coll_wall) { r = -0.25; A.st |= CAGYM_FLAG_IN_COLLISION; }
                    else if (dmin <= 0.2) r += -0.1 - dmin / 2.;
                } else if (A.st & CAGYM_FLAG_RAN_OUT_OF_TIME) {
                    r += -10.0;
                }
            }
            r = clipd(r, -10.0, 3.0) / (3.0 - (-10.0));
            reward = (float)r;
            if (A.st & (CAGYM_FLAG_AT_GOAL | CAGYM_FLAG_RAN_OUT_OF_TIME | CAGYM_FLAG_IN_COLLISION)) A.st |= CAGYM_FLAG_DONE;
        }
        const bool live = C.valid && C.active;
        const bool done = !live || (A.st & CAGYM_FLAG_DONE);
        const uint64_t wm = world_mask64(C);
        const uint64_t b_done = __ballot(done);
        const uint64_t b_learn = __ballot(done || ST_POLICY(A.st) != CAGYM_POL_LEARNING);
        bool go;
        if (D.go_mode == CAGYM_GO_ALL) go = (b_done & wm) == wm;
        else if (D.go_mode == CAGYM_GO_LEARNING) go = (b_learn & wm) == wm;
        else go = C.n > 0 ? ((b_done >> C.base) & 1ull) : true;
        if (C.valid) {
            if (out.reward) out.reward[aidx] = reward;
            if (out.flags) out.flags[aidx] = (uint8_t)(A.st & 0xffu);
            if (C.slot == 0) {
                if (out.game_over) out.game_over[C.world] = go ? 1 : 0;
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
                fold_episode_stats(D, Cr, A, r0, l0);
                if (rs) {
                    ep_ret = 0.f;
                    ep_len = 0;
                    C.episode += 1;
                    int sidx = (int)(((long long)C.world + (long long)C.episode * D.N) % D.S);
                    C.n = D.sc_nagents[sidx];
                    C.active = C.slot < C.n;
                    init_agent(D, A, sidx, C.slot, C.active);
                    lds_store_agent(W, A, tid);
                    if (C.slot == 0) W.wn[C.wl] = C.n;
                }
            }
        }
        W.tst[tid] = A.st;
        if (tid == 0) W.flag[0] = any_reset ? 1 : 0;
    }
    __syncthreads();
    STAMP(5);
    // ---- P3: OtherAgentsStatesSensor rows (sensors/OtherAgentsStatesSensor.py:11-77) ----------------------
    if (AUTO_RESET && W.flag[0]) {  // rare: some world restarted -> sort keys of the new episode
        for (int p = tid; p < npairs; p += NT) {
            const PairIdx q = pair_of(p, M, inv_m);
            const int n = W.wn[q.wl];
            double key = -INFINITY;
            if (q.sl < n && q.j < n && q.j != q.sl) {
                const int b = q.a - q.sl + q.j;
                key = norm2(W.tpx[b] - W.tpx[q.a], W.tpy[b] - W.tpy[q.a]) - W.tr[q.a] - W.tr[b];
            }
            W.keys[q.a * MP + q.j] = key;
        }
        __syncthreads();
    }
    // every (agent, other slot) lane owns exactly one 40-B row of the agent's [K, 10] table and stores it straight to
    // HBM (five 8-B stores; the workgroup's rows are one contiguous 3600 B x worlds block, merged in L2)
    if (out.obs_oas)
    for (int p = tid; p < npairs; p += NT) {
        const PairIdx q = pair_of(p, M, inv_m);
        if (q.j == q.sl || q.wl >= C.worlds_valid) continue;
        const int n = W.wn[q.wl];
        float* my = out.obs_oas + ((size_t)blockIdx.x * C.wpw * M + q.a) * K * 10;
        const double kj = W.keys[q.a * MP + q.j];
        float v[10];
        int row;
        if (!(kj > -INFINITY)) {  // unused row: zero (rows n-1 .. K-1 of an active agent, every row of an empty slot)
            row = q.sl < n ? q.j - 1 : (q.j < q.sl ? q.j : q.j - 1);
#pragma unroll
            for (int c = 0; c < 10; c++) v[c] = 0.f;
        } else {
            int before = 0;  // descending key, ties by descending index (stable sort, reversed: :28-34)
            const double2* krow = reinterpret_cast<const double2*>(W.keys + q.a * MP);
            for (int l2 = 0; l2 < MP; l2 += 2) {
                const double2 kk = krow[l2 >> 1];
                before += (kk.x > kj) || (kk.x == kj && l2 + 0 > q.j);
                before += (kk.y > kj) || (kk.y == kj && l2 + 1 > q.j);
            }
            row = before;
            const int b = q.a - q.sl + q.j;
            const double dx = W.tpx[b] - W.tpx[q.a], dy = W.tpy[b] - W.tpy[q.a];
            const double prx = W.tprx[q.a], pry = W.tpry[q.a], orx = -pry, ory = prx;
            const double ovx = W.tvx[b], ovy = W.tvy[b], orad = W.tr[b];
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
        }
        float2* r2 = reinterpret_cast<float2*>(my + row * 10);  // rows are 40 B: 8-byte aligned
#pragma unroll
        for (int c = 0; c < 5; c++) r2[c] = make_float2(v[2 * c], v[2 * c + 1]);
    }
    // the last wave is idle in the second round of the row loop: it prepares the next step's LP inputs
    if (any_rvo && tid >= NT - CAGYM_WAVE && tid - (NT - CAGYM_WAVE) < C.wpw * M) publish_pref_velocity(W, tid - (NT - CAGYM_WAVE));
    STAMP(6);
    // ---- ego observation from the agent lanes ---------------------------------------------------------
    if (agent_lane && C.valid) {
        const int nobs = C.active ? C.n - 1 : 0;
        D.n_observed[aidx] = nobs;
        if (out.obs_ego) {
            float4* e = reinterpret_cast<float4*>(out.obs_ego + aidx * CAGYM_EGO_WIDTH);
            if (C.active) {
                const Agent A = lds_load_agent(W, tid);
                e[0] = make_float4((float)A.dg, (float)(A.gx - A.px), (float)(A.gy - A.py), (float)A.r);
                e[1] = make_float4((float)A.he, (float)A.h, (float)A.px, (float)A.py);
                e[2] = make_float4((float)A.pref, (float)nobs, ST_POLICY(A.st) == CAGYM_POL_LEARNING ? 1.f : 0.f, 0.f);
            } else {
                e[0] = e[1] = e[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    STAMP(7);
    // the next S0 barrier orders these staging reads before the union is rewritten
}

template <int NT, int MT, int WPWT, bool AUTO_RESET>
__global__ void __launch_bounds__(NT, (NT <= 256 ? 3 : 2)) k_rollout2(CagymDev D, int n_steps, CagymOut out, int any_rvo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WGTRACE(0);
#ifdef CAGYM_WGTRACE
    if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG) g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + 39] = 0;
#endif
    const int M = MT ? MT : D.M;
    Lds2 W = carve_lds2(smem, M, cagym_as(M, WPWT));
    LaneCtx C = make_ctx2(D, M, WPWT ? WPWT : CAGYM_WAVE / M);
    const bool agent_lane = (int)threadIdx.x < C.wpw * M;
    const size_t aidx = (size_t)C.world * M + C.slot;
    float ep_ret = 0.f;
    int ep_len = 0;
    if (!agent_lane) C.valid = C.active = false;
    if (agent_lane) {
        Agent A = {};
        if (C.valid) {
            load_agent(D, A, aidx);
            if (C.slot == 0) { ep_ret = D.ep_return[C.world]; ep_len = D.ep_len[C.world]; }
        }
        lds_store_agent(W, A, threadIdx.x);
        if (C.wl < C.wpw && C.slot == 0) W.wn[C.wl] = C.valid ? C.n : 0;
        publish_pref_velocity(W, threadIdx.x);
    }
    WGTRACE(1);
    const size_t NM = (size_t)D.N * M;
#pragma nounroll
    for (int t = 0; t < n_steps; t++) {
        CagymOut o;
        o.obs_oas = out.obs_oas ? out.obs_oas + (size_t)t * NM * (M - 1) * 10 : nullptr;
        o.obs_ego = out.obs_ego ? out.obs_ego + (size_t)t * NM * CAGYM_EGO_WIDTH : nullptr;
        o.laserscan = nullptr;
        o.reward = out.reward ? out.reward + (size_t)t * NM : nullptr;
        o.flags = out.flags ? out.flags + (size_t)t * NM : nullptr;
        o.game_over = out.game_over ? out.game_over + (size_t)t * D.N : nullptr;
        step_core2<NT, MT, WPWT, AUTO_RESET>(D, W, C, nullptr, o, ep_ret, ep_len, any_rvo != 0);
        WGTRACE(2 + t);
    }
    if (C.valid) {
        const Agent A = lds_load_agent(W, threadIdx.x);  // own lane's record: no barrier needed
        store_agent(D, A, aidx, true);
        if (C.slot == 0) {
            D.ep_return[C.world] = ep_ret;
            D.ep_len[C.world] = ep_len;
            D.episode[C.world] = C.episode;
            D.n_agents[C.world] = C.n;
        }
    }
#ifdef CAGYM_WGTRACE
    WGTRACE(38);
    if (threadIdx.x == 0 && blockIdx.x < CAGYM_WGTRACE_MAXWG)
        g_wgtrace[blockIdx.x * CAGYM_WGTRACE_W + 39] |= __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u;  // HW_REG_XCC_ID
#endif
}

template <int NT, int MT, int WPWT, bool AUTO_RESET>
__global__ void __launch_bounds__(NT) k_step2(CagymDev D, const float* ext, CagymOut out, int any_rvo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = MT ? MT : D.M;
    Lds2 W = carve_lds2(smem, M, cagym_as(M, WPWT));
    LaneCtx C = make_ctx2(D, M, WPWT ? WPWT : CAGYM_WAVE / M);
    const bool agent_lane = (int)threadIdx.x < C.wpw * M;
    const size_t aidx = (size_t)C.world * M + C.slot;
    float ep_ret = 0.f;
    int ep_len = 0;
    if (!agent_lane) C.valid = C.active = false;
    if (agent_lane) {
        Agent A = {};
        if (C.valid) {
            load_agent(D, A, aidx);
            if (C.slot == 0) { ep_ret = D.ep_return[C.world]; ep_len = D.ep_len[C.world]; }
        }
        lds_store_agent(W, A, threadIdx.x);
        if (C.wl < C.wpw && C.slot == 0) W.wn[C.wl] = C.valid ? C.n : 0;
        publish_pref_velocity(W, threadIdx.x);
    }
    step_core2<NT, MT, WPWT, AUTO_RESET>(D, W, C, ext, out, ep_ret, ep_len, any_rvo != 0);
    if (C.valid) {
        const Agent A = lds_load_agent(W, threadIdx.x);
        store_agent(D, A, aidx, AUTO_RESET);
        if (C.slot == 0) {
            D.ep_return[C.world] = ep_ret;
            D.ep_len[C.world] = ep_len;
            if (AUTO_RESET) {
                D.episode[C.world] = C.episode;
                D.n_agents[C.world] = C.n;
            }
        }
    }
}
