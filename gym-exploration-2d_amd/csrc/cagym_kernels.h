// cagym_kernels.h -- the fused env.step() kernel family (gfx950).
//
// Launch geometry: one 64-lane wavefront per workgroup; a wave owns wpw = 64 / M whole worlds
// (lane = world_in_wave * M + slot).  A 4096 x 10 batch is only 683 waves on a 1024-SIMD chip, so
// the launch is latency-bound by construction: single-wave workgroups spread over all CUs/XCDs
// and need no s_barrier between phases other than the wave-level LDS ordering.
//
// HBM traffic per agent-step (single step): ~170 B state in, ~130 B state out, 360 B OAS + 48 B
// ego obs + 5 B reward/flags out; the OAS table is staged in LDS and written as contiguous 16-B
// lanes (1 KiB per wave store).  In the rollout kernel the state stays in registers across steps.
#pragma once
#include "cagym_device.h"
#include "cagym_orca.h"

// LDS carve for one wave.
struct WaveLds {
    double *tpx, *tpy, *tvx, *tvy, *tr;  // neighbour tile [64]
    uint32_t* tst;                        // status words [64]
    double* keys;                         // OAS sort keys [(M-1)][64]
    float* oas;                           // OAS staging [64*(M-1)*10] floats   } union
    float4* lines;                        // ORCA lines [2][M - 1][64]           }
};

// ORCA line arrays of generation 1: [2][M - 1][64] (every other agent may be a neighbour: maxNeighbors <= M - 1)
__host__ __device__ inline size_t cagym_lds_bytes(int M) {
    size_t tile = 5 * 64 * 8 + 64 * 4;
    size_t keys = (size_t)(M - 1) * 64 * 8;
    size_t oas = (size_t)64 * (M - 1) * 10 * 4;
    size_t lines = (size_t)2 * (M - 1) * 64 * 16;
    return tile + keys + (oas > lines ? oas : lines);
}

__device__ __forceinline__ WaveLds carve_lds(unsigned char* smem, int M) {
    WaveLds W;
    W.tpx = reinterpret_cast<double*>(smem);
    W.tpy = W.tpx + 64;
    W.tvx = W.tpy + 64;
    W.tvy = W.tvx + 64;
    W.tr = W.tvy + 64;
    W.keys = W.tr + 64;
    unsigned char* p = reinterpret_cast<unsigned char*>(W.keys + (size_t)(M - 1) * 64);
    W.oas = reinterpret_cast<float*>(p);
    W.lines = reinterpret_cast<float4*>(p);
    size_t oas = (size_t)64 * (M - 1) * 10 * 4, lines = (size_t)2 * (M - 1) * 64 * 16;
    W.tst = reinterpret_cast<uint32_t*>(p + (oas > lines ? oas : lines));
    return W;
}

// One wave = one workgroup, so the workgroup barrier is the wave-level LDS ordering point.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

struct LaneCtx {
    int lane, wl, slot, base, world, n, wpw, worlds_valid;
    int episode;  // episodes started by this world (every lane of the world tracks it)
    bool valid;   // lane maps to an existing world
    bool active;  // slot < n_agents[world]
};

__device__ __forceinline__ void publish_tile(const WaveLds& W, const Agent& A, int lane) {
    W.tpx[lane] = A.px;
    W.tpy[lane] = A.py;
    W.tvx[lane] = A.vx;
    W.tvy[lane] = A.vy;
    W.tr[lane] = A.r;
    W.tst[lane] = A.st;
}

__device__ __forceinline__ uint64_t world_mask64(const LaneCtx& C) {
    uint64_t m = C.n >= 64 ? ~0ull : ((1ull << C.n) - 1ull);
    return m << C.base;
}

// fold the finished episode of this lane's world into the cumulative statistics (lane slot 0 writes)
__device__ __forceinline__ void fold_episode_stats(const CagymDev& D, const LaneCtx& C, const Agent& A, float& ep_ret,
                                                   int& ep_len) {
    uint64_t wm = world_mask64(C);
    bool live = C.valid && C.active;
    int goal = __popcll(__ballot(live && (A.st & CAGYM_FLAG_AT_GOAL)) & wm);
    int coll = __popcll(__ballot(live && (A.st & CAGYM_FLAG_IN_COLLISION)) & wm);
    int tout = __popcll(__ballot(live && (A.st & CAGYM_FLAG_RAN_OUT_OF_TIME)) & wm);
    if (C.valid && C.slot == 0) {
        // every load before the first store: written as six "+=" the compiler must assume the arrays alias and waits for each
        // read-modify-write in turn - four dependent HBM round trips (~3.6 us per restarted world in the step kernels)
        const float r0 = D.stat_return[C.world];
        const int e0 = D.stat_episodes[C.world], s0 = D.stat_steps[C.world];
        const int o0 = D.stat_outcomes[C.world * 3 + 0], o1 = D.stat_outcomes[C.world * 3 + 1], o2 = D.stat_outcomes[C.world * 3 + 2];
        D.stat_return[C.world] = r0 + ep_ret;
        D.stat_episodes[C.world] = e0 + 1;
        D.stat_steps[C.world] = s0 + ep_len;
        D.stat_outcomes[C.world * 3 + 0] = o0 + goal;
        D.stat_outcomes[C.world * 3 + 1] = o1 + coll;
        D.stat_outcomes[C.world * 3 + 2] = o2 + tout;
    }
    ep_ret = 0.f;
    ep_len = 0;
}

// OtherAgentsStatesSensor.sense (sensors/OtherAgentsStatesSensor.py:11-77) + the scalar observation
// keys (agent.py:244-248, config.py:104-215).  Tile must hold the CURRENT positions/velocities.
// write_mask: bit wl set => world wl of this wave is written to `out`.
__device__ inline void sense_and_store(const CagymDev& D, const WaveLds& W, const LaneCtx& C, Agent& A,
                                       const CagymOut& out, uint64_t write_worlds, bool all_worlds) {
    const int M = D.M, K = M - 1, lane = C.lane;
    int nobs = 0;
    if (C.valid) {
        float* my = W.oas + (size_t)lane * K * 10;
        for (int q = 0; q < K * 10; q++) my[q] = 0.f;
        if (C.active) {
            double prx, pry;
            ref_axes(A, prx, pry);
            const double orx = -pry, ory = prx;
            int cnt = 0;
            for (int j = 0; j < C.n; j++) {
                if (j == C.slot) continue;
                double dx = W.tpx[C.base + j] - A.px, dy = W.tpy[C.base + j] - A.py;
                W.keys[cnt * 64 + lane] = norm2(dx, dy) - A.r - W.tr[C.base + j];
                cnt++;
            }
            // stable ascending sort, reversed, last K kept (:28-34)  ==  descending key, ties by
            // descending index; row = (#others ranked before) - (cnt - kept)
            const int kept = cnt < K ? cnt : K;
            const int drop = cnt - kept;
            int c1 = 0;
            for (int j = 0; j < C.n; j++) {
                if (j == C.slot) continue;
                double kj = W.keys[c1 * 64 + lane];
                int before = 0, c2 = 0;
                for (int l = 0; l < C.n; l++) {
                    if (l == C.slot) continue;
                    double kl = W.keys[c2 * 64 + lane];
                    before += (kl > kj) || (kl == kj && l > j);
                    c2++;
                }
                c1++;
                int row = before - drop;
                if (row < 0) continue;
                double dx = W.tpx[C.base + j] - A.px, dy = W.tpy[C.base + j] - A.py;
                double ovx = W.tvx[C.base + j], ovy = W.tvy[C.base + j], orad = W.tr[C.base + j];
                float* r = my + row * 10;
                r[0] = (float)dx;
                r[1] = (float)dy;
                r[2] = (float)dot2(dx, dy, prx, pry);
                r[3] = (float)dot2(dx, dy, orx, ory);
                r[4] = (float)dot2(ovx, ovy, prx, pry);
                r[5] = (float)dot2(ovx, ovy, orx, ory);
                r[6] = (float)orad;
                r[7] = (float)(A.r + orad);
                r[8] = (float)kj;
                r[9] = ST_POLICY(W.tst[C.base + j]) == CAGYM_POL_STATIC ? 1.f : 2.f;
            }
            nobs = kept;
        }
        D.n_observed[(size_t)C.world * M + C.slot] = nobs;
    }
    wave_sync();
    // coalesced store of the staged table
    if (out.obs_oas) {
        const size_t world0 = (size_t)(blockIdx.x) * C.wpw;
        const int per_world4 = M * K * 10 / 4;
        const float4* src = reinterpret_cast<const float4*>(W.oas);
        float4* dst = reinterpret_cast<float4*>(out.obs_oas) + world0 * per_world4;
        if (all_worlds) {
            const int total4 = C.worlds_valid * per_world4;
            for (int q = lane; q < total4; q += 64) dst[q] = src[q];
        } else {
            for (int wl = 0; wl < C.worlds_valid; wl++) {
                if (!((write_worlds >> wl) & 1ull)) continue;
                for (int q = lane; q < per_world4; q += 64) dst[wl * per_world4 + q] = src[wl * per_world4 + q];
            }
        }
    }
    bool wr = C.valid && (all_worlds || ((write_worlds >> C.wl) & 1ull));
    if (out.obs_ego && wr) {
        float4* e = reinterpret_cast<float4*>(out.obs_ego + ((size_t)C.world * M + C.slot) * CAGYM_EGO_WIDTH);
        if (C.active) {
            e[0] = make_float4((float)A.dg, (float)(A.gx - A.px), (float)(A.gy - A.py), (float)A.r);
            e[1] = make_float4((float)A.he, (float)A.h, (float)A.px, (float)A.py);
            e[2] = make_float4((float)A.pref, (float)nobs, ST_POLICY(A.st) == CAGYM_POL_LEARNING ? 1.f : 0.f, 0.f);
        } else {
            e[0] = e[1] = e[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    wave_sync();  // staging region is free again
}

// One env.step() for the worlds of this wave (env.py:162-232).  A: register state of this lane.
// ext: external action pairs [N,M,2] or null.  out: output slice of this step.
template <bool AUTO_RESET>
__device__ inline void step_core(const CagymDev& D, const WaveLds& W, LaneCtx& C, Agent& A, const float* ext,
                                 const CagymOut& out, float& ep_ret, int& ep_len) {
    const int lane = C.lane, M = D.M;
    const size_t aidx = (size_t)C.world * M + C.slot;
    // ---- _take_action (env.py:287-340): all agents select, then all move --------------------
    publish_tile(W, A, lane);
    wave_sync();
    float a0 = 0.f, a1 = 0.f;
    HeadingHint hint;
    hint.valid = false;
    if (C.valid && C.active && !(A.st & CAGYM_FLAG_DONE)) {
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
                NbrTile T{W.tpx, W.tpy, W.tvx, W.tvy, W.tr};
                orca_action(T, W.lines, W.lines + (M - 1) * 64, lane, C.base, C.n, C.slot, A, D.dt, D.maxnb, d0, d1, &hint);
                break;
            }
        }
        a0 = (float)d0;
        a1 = (float)d1;
    }
    if (C.valid && C.active) take_action(A, a0, a1, D.dt, &hint);
    wave_sync();  // every lane is done reading the pre-move tile
    publish_tile(W, A, lane);
    wave_sync();
    // ---- _compute_rewards + _check_for_collisions (env.py:502-567, 630-671) -------------------
    float reward = 0.f;
    if (C.valid && C.active) {
        bool coll_agent = false, coll_wall = false;
        double dmin = INFINITY;
        const bool self_static = ST_POLICY(A.st) == CAGYM_POL_STATIC;
        for (int j = 0; j < C.n; j++) {
            if (j == C.slot) continue;
            const bool other_static = ST_POLICY(W.tst[C.base + j]) == CAGYM_POL_STATIC;
            // pair (lo, hi): skipped when agent hi is Static (env.py:643, Q8)
            const bool skip = ((j > C.slot) ? other_static : self_static) && !D.collide_static;
            if (skip) continue;
            double d = norm2(A.px - W.tpx[C.base + j], A.py - W.tpy[C.base + j]);
            double cr = (j > C.slot) ? (A.r + W.tr[C.base + j]) : (W.tr[C.base + j] + A.r);
            if (j > C.slot) {  // dist_btwn_nearest_agent is only updated for the lower index (Q7)
                double g = d - cr;
                if (g < dmin) dmin = g;
            }
            if (d <= cr) coll_agent = true;
        }
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
                r += -10.0;  // Q9
            }
        }
        r = clipd(r, -10.0, 3.0) / (3.0 - (-10.0));
        reward = (float)r;
        // ---- _check_which_agents_done (env.py:711-721) --------------------------------------
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
    if (AUTO_RESET) {
        // DummyVecEnv semantics (exp/env_utils.py:29-31): the finished world restarts on its next
        // scenario and the observation returned for this step is the first one of the new episode.
        const bool rs = C.valid && go;
        if (__ballot(rs)) {
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
            }
        }
        wave_sync();
        publish_tile(W, A, lane);
        wave_sync();
    }
    // ---- _get_obs (env.py:740-753) ----------------------------------------------------------
    sense_and_store(D, W, C, A, out, 0, true);
}

__device__ __forceinline__ LaneCtx make_ctx(const CagymDev& D) {
    LaneCtx C;
    C.lane = threadIdx.x;
    C.wpw = CAGYM_WAVE / D.M;
    C.wl = C.lane / D.M;
    C.slot = C.lane - C.wl * D.M;
    C.base = C.wl * D.M;
    C.world = blockIdx.x * C.wpw + C.wl;
    int rem = D.N - (int)blockIdx.x * C.wpw;
    C.worlds_valid = rem < C.wpw ? rem : C.wpw;
    C.valid = C.wl < C.wpw && C.world < D.N;
    C.n = C.valid ? D.n_agents[C.world] : 0;
    C.episode = C.valid ? D.episode[C.world] : 0;
    C.active = C.valid && C.slot < C.n;
    return C;
}

// env.step(): one launch per step (external actions allowed).
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
__global__ void __launch_bounds__(64) k_step(CagymDev D, const float* ext, CagymOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds W = carve_lds(smem, D.M);
    LaneCtx C = make_ctx(D);
    Agent A = {};
    const size_t aidx = (size_t)C.world * D.M + C.slot;
    float ep_ret = 0.f;
    int ep_len = 0;
    if (C.valid) {
        load_agent(D, A, aidx);
        if (C.slot == 0) { ep_ret = D.ep_return[C.world]; ep_len = D.ep_len[C.world]; }
    }
    step_core<false>(D, W, C, A, ext, out, ep_ret, ep_len);
    if (C.valid) {
        store_agent(D, A, aidx, false);
        if (C.slot == 0) { D.ep_return[C.world] = ep_ret; D.ep_len[C.world] = ep_len; }
    }
}
#endif

// n_steps env.step() calls in one launch, state in registers, outputs to slice t (cagym_rollout).
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
template <bool AUTO_RESET>
__global__ void __launch_bounds__(64) k_rollout(CagymDev D, int n_steps, CagymOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds W = carve_lds(smem, D.M);
    LaneCtx C = make_ctx(D);
    Agent A = {};
    size_t aidx = (size_t)C.world * D.M + C.slot;
    float ep_ret = 0.f;
    int ep_len = 0;
    if (C.valid) {
        load_agent(D, A, aidx);
        if (C.slot == 0) { ep_ret = D.ep_return[C.world]; ep_len = D.ep_len[C.world]; }
    }
    const size_t NM = (size_t)D.N * D.M;
    for (int t = 0; t < n_steps; t++) {
        CagymOut o;
        o.obs_oas = out.obs_oas ? out.obs_oas + (size_t)t * NM * (D.M - 1) * 10 : nullptr;
        o.obs_ego = out.obs_ego ? out.obs_ego + (size_t)t * NM * CAGYM_EGO_WIDTH : nullptr;
        o.laserscan = nullptr;
        o.reward = out.reward ? out.reward + (size_t)t * NM : nullptr;
        o.flags = out.flags ? out.flags + (size_t)t * NM : nullptr;
        o.game_over = out.game_over ? out.game_over + (size_t)t * D.N : nullptr;
        step_core<AUTO_RESET>(D, W, C, A, nullptr, o, ep_ret, ep_len);
    }
    if (C.valid) {
        store_agent(D, A, aidx, true);
        if (C.slot == 0) {
            D.ep_return[C.world] = ep_ret;
            D.ep_len[C.world] = ep_len;
            D.episode[C.world] = C.episode;
            D.n_agents[C.world] = C.n;
        }
    }
}
#endif

// reset() (env.py:234-266) for masked worlds.
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
__global__ void __launch_bounds__(64) k_reset(CagymDev D, const uint8_t* mask, int advance, CagymOut out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds W = carve_lds(smem, D.M);
    LaneCtx C = make_ctx(D);
    Agent A = {};
    const size_t aidx = (size_t)C.world * D.M + C.slot;
    const bool rs = C.valid && (!mask || mask[C.world]);
    if (C.valid) load_agent(D, A, aidx);
    if (advance) {
        float r0 = 0.f;
        int l0 = 0;
        if (rs && C.slot == 0) { r0 = D.ep_return[C.world]; l0 = D.ep_len[C.world]; }
        LaneCtx Cr = C;
        Cr.valid = rs;
        fold_episode_stats(D, Cr, A, r0, l0);
    }
    if (rs) {
        if (advance) C.episode += 1;
        int sidx = (int)(((long long)C.world + (long long)C.episode * D.N) % D.S);
        C.n = D.sc_nagents[sidx];
        C.active = C.slot < C.n;
        init_agent(D, A, sidx, C.slot, C.active);
    }
    wave_sync();
    if (rs) {
        store_agent(D, A, aidx, true);
        if (C.slot == 0) {
            D.episode[C.world] = C.episode;
            D.n_agents[C.world] = C.n;
            D.ep_return[C.world] = 0.f;
            D.ep_len[C.world] = 0;
            if (out.game_over) out.game_over[C.world] = 0;
        }
        if (out.reward) out.reward[aidx] = 0.f;
        if (out.flags) out.flags[aidx] = (uint8_t)(A.st & 0xffu);
    }
    publish_tile(W, A, C.lane);
    wave_sync();
    uint64_t wr = 0;
    {
        uint64_t b = __ballot(rs && C.slot == 0);
        for (int wl = 0; wl < C.wpw; wl++)
            if ((b >> (wl * D.M)) & 1ull) wr |= 1ull << wl;
    }
    sense_and_store(D, W, C, A, out, wr, false);
}
#endif

// LaserScanSensor.sense (sensors/LaserScanSensor.py:9-22,27-58), beam b of an agent at (px, py, heading h) with
// `radius`: 16 samples at 2 pi / 16 m into the bit-packed raster `map` (null = empty map), the agent's own disk masked,
// the LAST sample whose running hit count is 1 gives the range.
__device__ __forceinline__ float laserscan_beam(const uint32_t* map, double px, double py, double h, double radius, int b) {
    int egx, egy;
    const bool ego_in = world_to_cell(px, py, egx, egy);
    const double rr = radius / 0.1, r2 = rr * rr;
    const double astep = (kPi - (-kPi)) / 15.0, rstep = 2 * kPi / 16;
    const double ang0 = b == 15 ? kPi : (double)b * astep + (-kPi);
    double sa, ca;
    sincos(ang0 + h, &sa, &ca);
    // all 16 raster words of the beam are requested before the first one is looked at (16 gathers in flight instead of
    // one after the other: the raster is L2-resident, the latency is what costs)
    uint32_t word[16];
    int bit[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        double rg = 0.0 + (double)k * rstep;
        double x = px + rg * ca, y = py + rg * sa;
        int gx, gy;
        bool in = map && world_to_cell(x, y, gx, gy);
        bool masked = false;
        if (in && ego_in) {
            double dx = (double)(gy - egy), dy = (double)(gx - egx);
            masked = dx * dx + dy * dy < r2;
        }
        in = in && !masked;
        bit[k] = in ? (gy & 31) : -1;
        word[k] = in ? map[gx * CAGYM_MAPW + (gy >> 5)] : 0u;
    }
    int count = 0, last = -1;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const bool hit = bit[k] >= 0 && ((word[k] >> (bit[k] & 31)) & 1u);
        count += hit ? 1 : 0;
        if (count == 1) last = k;
    }
    double range = last >= 0 ? 0.0 + (double)last * rstep : 6.0;
    return (float)(1 - range / 6);
}

// one lane per (agent, beam) of the state in HBM (cagym_laserscan, cagym_reset)
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
__global__ void __launch_bounds__(256) k_laserscan(CagymDev D, float* out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)D.N * D.M * 16;
    if (gid >= total) return;
    const size_t a = gid >> 4;
    const int b = (int)(gid & 15);
    const int world = (int)(a / D.M), slot = (int)(a - (size_t)world * D.M);
    if (slot >= D.n_agents[world]) { out[gid] = 0.f; return; }
    const int sidx = (int)(((long long)world + (long long)D.episode[world] * D.N) % D.S);
    const uint32_t* map = (D.map_bits && D.sc_nobst[sidx] > 0) ? D.map_bits + (size_t)sidx * CAGYM_MAPD * CAGYM_MAPW : nullptr;
    out[gid] = laserscan_beam(map, D.px[a], D.py[a], D.heading[a], D.radius[a], b);
}
#endif

// OccupancyGridSensor.sense (sensors/OccupancyGridSensor.py:70-98, 131-143; Map.getSubmapByIndices Map.py:81-105):
// the occupancy raster rotated about the agent's cell by -heading (cv2.getRotationMatrix2D + cv2.warpAffine,
// bilinear, constant-0 border), 60x60 window around the agent, astype(bool).  One workgroup per agent, 3600 cells.
// warpAffine restated from OpenCV 4.x imgproc (fixed point: AB_BITS 10, INTER_BITS 5, round_delta 16, cvRound =
// round half to even); cv2 is absent here, so this sensor is "parity unpinned" (oracle twin: cagym_oracle_grid.c).
__device__ __forceinline__ int og_src(const uint32_t* map, int x, int y) {  // x = column, y = row; border 0
    if (x < 0 || y < 0 || x >= CAGYM_MAPD || y >= CAGYM_MAPD) return 0;
    return map_bit(map, y, x) ? 1 : 0;
}
__device__ __forceinline__ int og_submap_start(int c) {
    long long si = (long long)((double)c - floor(60 / 2.0));  // int(): truncation toward zero
    if (si < 0) si = 0;
    if (si + 60 > CAGYM_MAPD - 1) si = CAGYM_MAPD - 1 - 60;
    return (int)si;
}
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
__global__ void __launch_bounds__(256) k_occupancy_grid(CagymDev D, uint8_t* out) {
    const size_t a = blockIdx.x;
    const int world = (int)(a / D.M), slot = (int)(a - (size_t)world * D.M);
    uint8_t* o = out + a * 3600;
    const int sidx = (int)(((long long)world + (long long)D.episode[world] * D.N) % D.S);
    const bool live = slot < D.n_agents[world] && D.map_bits && D.sc_nobst[sidx] > 0;
    if (!live) {  // inactive slot or empty map: all free
        for (int q = threadIdx.x; q < 3600; q += blockDim.x) o[q] = 0;
        return;
    }
    const uint32_t* map = D.map_bits + (size_t)sidx * CAGYM_MAPD * CAGYM_MAPW;
    int gx, gy;
    world_to_cell(D.px[a], D.py[a], gx, gy);
    const int sx0 = og_submap_start(gx), sy0 = og_submap_start(gy);
    double angle = -D.heading[a] * 180 / kPi;
    angle *= kPi / 180;
    double beta, alpha;
    sincos(angle, &beta, &alpha);
    const double cx = (double)gy, cy = (double)gx;
    double M0 = alpha, M1 = beta, M2 = (1 - alpha) * cx - beta * cy, M3 = -beta, M4 = alpha, M5 = beta * cx + (1 - alpha) * cy;
    double Dt = M0 * M4 - M1 * M3;
    Dt = Dt != 0 ? 1. / Dt : 0;
    const double A11 = M4 * Dt, A22 = M0 * Dt;
    M0 = A11; M1 *= -Dt;
    M3 *= -Dt; M4 = A22;
    const double b1 = -M0 * M2 - M1 * M5;
    const double b2 = -M3 * M2 - M4 * M5;
    M2 = b1; M5 = b2;
    for (int q = threadIdx.x; q < 3600; q += blockDim.x) {
        const int r = q / 60, c = q - r * 60;
        const int y = sx0 + r, x = sy0 + c;
        const int X0 = __double2int_rn((M1 * y + M2) * 1024) + 16;
        const int Y0 = __double2int_rn((M4 * y + M5) * 1024) + 16;
        const int X = (X0 + __double2int_rn(M0 * x * 1024)) >> 5;
        const int Y = (Y0 + __double2int_rn(M3 * x * 1024)) >> 5;
        const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
        int v = og_src(map, sx, sy);
        if (fx) v |= og_src(map, sx + 1, sy);
        if (fy) v |= og_src(map, sx, sy + 1);
        if (fx && fy) v |= og_src(map, sx + 1, sy + 1);
        o[q] = (uint8_t)v;
    }
}
#endif

// Map.get_occupancy_grid (Map.py:107-123): one workgroup per scenario, bit-packed output.
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
__global__ void __launch_bounds__(256) k_rasterize(const double* obst, const int32_t* nobst, int Kobs, uint32_t* map_bits) {
    const int s = blockIdx.x;
    uint32_t* map = map_bits + (size_t)s * CAGYM_MAPD * CAGYM_MAPW;
    for (int q = threadIdx.x; q < CAGYM_MAPD * CAGYM_MAPW; q += blockDim.x) map[q] = 0u;
    __syncthreads();
    const int n = nobst[s];
    for (int o = 0; o < n; o++) {
        const double* ob = obst + ((size_t)s * Kobs + o) * 4;
        int s0, s1, e0, e1;
        world_to_cell(ob[0], ob[3], s0, s1);  // corner[1] = (xl, yu)
        world_to_cell(ob[2], ob[1], e0, e1);  // corner[3] = (xu, yl)
        if (s0 < -CAGYM_MAPD) s0 = -CAGYM_MAPD;
        if (s1 < -CAGYM_MAPD) s1 = -CAGYM_MAPD;
        if (e0 > CAGYM_MAPD - 1) e0 = CAGYM_MAPD - 1;
        if (e1 > CAGYM_MAPD - 1) e1 = CAGYM_MAPD - 1;
        const int h = e0 - s0 + 1, w = e1 - s1 + 1;
        if (h <= 0 || w <= 0) continue;
        for (int q = threadIdx.x; q < h * w; q += blockDim.x) {
            int ii = s0 + q / w, jj = s1 + q % w;
            int a = ii < 0 ? ii + CAGYM_MAPD : ii, b = jj < 0 ? jj + CAGYM_MAPD : jj;  // Python negative-index wrap
            if (a < 0 || b < 0 || a >= CAGYM_MAPD || b >= CAGYM_MAPD) continue;
            atomicOr(&map[a * CAGYM_MAPW + (b >> 5)], 1u << (b & 31));
        }
        __syncthreads();
    }
}
#endif

// GA3CCADRLPolicy.agents_to_ga3c_cadrl_state (policies/GA3CCADRLPolicy.py:45-106): LPA lanes per agent (lane j <-> other agent
// j: its distance, sort key and feature row; the rank is a count over the keys the agent's lanes left in LDS),
// out[N,M,76] f32 = [id, n_others, dist_to_goal, heading_ego, pref_speed, radius, 10 x 7 other-agent features], rows indexed by
// flat agent (world * M + slot) - by place in the list for cagym_ga3c_act -; others ordered by (-round(d,2), p_orth), stable, last `max_observed` kept.  Zero rows for
// inactive slots.  agent_idx == null: every agent slot of the handle; else the B (or *B_dev) listed agents only - the
// reference builds the vector for the GA3C agent alone (find_next_action is per agent).
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
// the state row of one agent on the LPA lanes of group `al` (every thread of the block calls it: one barrier inside)
template <int LPA>
__device__ __forceinline__ void ga3c_state_row(const CagymDev& D, int max_observed, bool have, size_t a, size_t orow, int al, int j,
                                               double (*sk1)[LPA], double (*sk2)[LPA], float* out) {
    const int world = (int)(a / D.M), i = (int)(a - (size_t)world * D.M);
    float* o = out + orow * 76;
    if (have)
        for (int c = j; c < 76; c += LPA) o[c] = 0.f;
    // every load the lane may need is requested before the first is looked at (slot indices clamped into the world: in bounds whatever
    // the world's agent count is) - the agent count, the ego's and the other agent's records come back in ONE round trip instead of three
    const size_t base = (size_t)world * D.M;
    const size_t oj = base + (size_t)(j < D.M ? j : D.M - 1);
    int n = 0;
    double px = 0, py = 0, ri = 0, gxa = 0, gya = 0, pxj = 0, pyj = 0, rj = 0, vxj = 0, vyj = 0;
    if (have) {
        n = D.n_agents[world];
        px = D.px[a]; py = D.py[a]; ri = D.radius[a]; gxa = D.gx[a]; gya = D.gy[a];
        pxj = D.px[oj]; pyj = D.py[oj]; rj = D.radius[oj]; vxj = D.vx[oj]; vyj = D.vy[oj];
    }
    const bool ego_live = have && i < n;
    double prx = 0, pry = 0, orx = 0, ory = 0;
    double dx = 0, dy = 0, ed = 0, k1 = 0, k2 = 0;
    const bool mine = ego_live && j < n && j != i;
    if (ego_live) {
        const double gx = gxa - px, gy = gya - py;
        const double dist = sqrt(gx * gx + gy * gy);
        prx = gx; pry = gy;
        if (dist > 1e-8) { prx = gx / dist; pry = gy / dist; }
        orx = -pry; ory = prx;
    }
    if (mine) {
        dx = pxj - px; dy = pyj - py;
        ed = norm2(dx, dy) - ri - rj;
        k1 = -(rint(ed * 100.0) / 100.0);
        k2 = dot2(dx, dy, orx, ory);
        sk1[al][j] = k1;
        sk2[al][j] = k2;
    }
    __syncthreads();  // keys of the agent's lanes; also orders the zero fill before the row stores below
    const int cnt = n - 1;
    const int drop = cnt > max_observed ? cnt - max_observed : 0;
    if (mine) {
        int before = 0;  // others sorted strictly before j: smaller (k1, k2), ties by lower index (stable)
        for (int l = 0; l < n; l++) {
            if (l == i || l == j) continue;
            const double l1 = sk1[al][l], l2 = sk2[al][l];
            before += (l1 < k1) || (l1 == k1 && (l2 < k2 || (l2 == k2 && l < j)));
        }
        const int row = before - drop;
        if (row >= 0) {
            const double vx = vxj, vy = vyj;
            float* r = o + 6 + 7 * row;
            r[0] = (float)dot2(dx, dy, prx, pry);
            r[1] = (float)k2;
            r[2] = (float)dot2(vx, vy, prx, pry);
            r[3] = (float)dot2(vx, vy, orx, ory);
            r[4] = (float)rj;
            r[5] = (float)(ri + rj);
            r[6] = (float)ed;
        }
    }
    if (ego_live && j == i) {
        o[0] = (float)i;
        o[1] = (float)(cnt - drop);  // rows kept: the ranks are a permutation of 0 .. cnt - 1
        o[2] = (float)D.dist_goal[a];
        o[3] = (float)D.heading_ego[a];
        o[4] = (float)D.pref[a];
        o[5] = (float)ri;
    }
}

template <int LPA>
__global__ void __launch_bounds__(256) k_ga3c_state(CagymDev D, int max_observed, const int32_t* __restrict__ agent_idx, int B,
                                                    uint32_t* ctr, float* out) {
    constexpr int APB = 256 / LPA;
    __shared__ double sk1[APB][LPA], sk2[APB][LPA];
    const int al = threadIdx.x / LPA, j = threadIdx.x % LPA;
    // ctr (cagym_ga3c_act): the list k_ga3c_select just built holds ctr[0] - ctr[1] agents; word 2 passes that on to the forward kernel
    const uint32_t listed = ctr ? ctr[0] - ctr[1] : 0u;
    if (ctr && blockIdx.x == 0 && threadIdx.x == 0) ctr[2] = listed;
    const long long total = agent_idx ? (long long)(ctr ? (int)listed : B) : (long long)D.N * D.M;
    if ((long long)blockIdx.x * APB >= total) return;  // uniform: the whole block is beyond the list
    const long long q = (long long)blockIdx.x * APB + al;
    const bool have = q < total;
    const size_t a = have ? (agent_idx ? (size_t)agent_idx[q] : (size_t)q) : 0;
    // cagym_ga3c_act's rows are stored by PLACE IN THE LIST (the forward kernel's tile of 32 agents is one contiguous 9.7 KB
    // block and needs no index look-up in front of its loads); every other caller gets rows indexed by flat agent
    ga3c_state_row<LPA>(D, max_observed, have, a, (ctr && have) ? (size_t)q : a, al, j, sk1, sk2, out);
}

#endif

// indices (world * M + slot) of the active agents whose policy id is CAGYM_POL_GA3C, compacted on the device (order within
// the list is not fixed: every consumer treats the listed agents independently).  The list needs no reset from the host
// (round 3; a memset in front of every call was 4.9 us of cfg4's step): ctr[0] is a ticket counter that only ever grows
// (unsigned, wraps), ctr[1] its value when this list began; a place in the list is ticket - ctr[1].  k_ga3c_state turns the
// difference into the list length (ctr[2]) and the forward kernel - the last reader - starts the next list (ctr[1] = ctr[0]).
// The words are the handle's, zero at creation; the chain replays from a captured graph as it is.  One call per handle in flight
// (include/cagym.h): a chain that was cut short (a failed launch) leaves the list open - cagym_ga3c_act re-zeroes the words then -
// and a place beyond the table (possible only then) is never stored.
#ifndef CAGYM_K3_UNIT  // the generation-3 units include this header for its device functions only
__global__ void __launch_bounds__(1024) k_ga3c_select(CagymDev D, int32_t* idx, uint32_t* ctr) {
    // one returning atomic per 1024-thread block (all of them hit one L2 address: per-wave atomics took 16 us for 81 920 slots)
    __shared__ int wave_cnt[16], wave_base[16];
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)D.N * D.M;
    bool take = false;
    if (a < total) {
        const uint32_t st = D.status[a];
        take = (st & CAGYM_FLAG_ACTIVE) && ST_POLICY(st) == CAGYM_POL_GA3C;
    }
    const unsigned long long m = __ballot(take);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; w++) { wave_base[w] = tot; tot += wave_cnt[w]; }
        const int base = tot ? (int)(atomicAdd(&ctr[0], (uint32_t)tot) - ctr[1]) : 0;
        for (int w = 0; w < 16; w++) wave_base[w] += base;
    }
    __syncthreads();
    if (take) {
        const size_t place = (size_t)(unsigned)(wave_base[wave] + __popcll(m & ((1ull << lane) - 1ull)));
        if (place < total) idx[place] = (int32_t)a;
    }
}
#endif
