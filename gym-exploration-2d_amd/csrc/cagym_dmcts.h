// cagym_dmcts.h -- Dec-MCTS planning step on the device (include/cagym.h: cagym_dmcts_plan).
//
// One 128-lane workgroup plans for all IG robots of one world: the UCT tree of every robot (selection, expansion,
// roll-outs, discounted back-propagation), the top-n action distribution and its exchange between the robots of the
// world.  Restates pydecmcts/DecMCTS.py:14-18, 92-231, 273-360 and ig_mcts.py:79-109 (reference paths under
// gym_collision_avoidance/envs/policies/); the executable specification is gym-exploration-2d_amd/dmcts.py
// (DecMCTSPlanner, host tree over the same device primitives): generator keys, summation orders and tie rules are
// the same, and the two make identical decisions (tests/test_dmcts.py).
//
// Tree bookkeeping is a few scalar operations per grow and runs on lane 0; what costs time -- the visibility set of a newly
// selected node, the Nsims x (horizon - stage) roll-out steps, the MI rewards, the top-n scan -- is spread over the workgroup
// with the functions of cagym_ig.h (so the numbers are those of cagym_ig_rollouts bit for bit).
//
// Round 3 (profiles/r3/cfg5_*): (1) the Nsims roll-outs of a grow are independent: each of the workgroup's 4 waves runs
// whole roll-outs on its own (visibility masks private to the wave, no workgroup barrier inside a roll-out; the MI sum keeps
// the summation order of the 128-thread block reduction, so every reward is the same double); (2) a node's "observed" mask
// (parent's mask | cells visible from its pose) is only ever read once the node has been SELECTED, so it is built then and
// not for each of the up to nine children at expansion - a quarter of the visibility queries of a grow - and the two 480-byte
// masks live in a pool with one entry per selected node: 80-byte nodes, 1.6 GB of workspace for 2048 worlds x 3 robots at
// the experiment's budget instead of 8.7 GB.
#pragma once
#include "cagym_ig.h"

#define DM_MAXH 8      // horizon
#define DM_MAXCOMM 8   // communicated plans per robot
#define DM_MAXR 8      // IG robots per world
#define DM_MAXSIMS 32
#ifndef DM_THREADS
#define DM_THREADS 128  /* lanes per world (round 4: 256 -> 128, see k_dmcts_plan) */
#endif
#define DM_WAVES (DM_THREADS / 64)
#define DM_WAVES_PER_SIMD 4  /* the register budget is held to 128 VGPRs: 16 waves per CU */
#define DM_VT 128      // virtual threads of the MI block reduction (ig_reward_block's order in cagym_ig_rollouts)
#define DM_NOACT 255   // no action stored
#define DM_INFEAS 254  // infeasible random draw: the roll-out appended (0, 0) (ig_mcts.py:226-231)

struct DmNode {  // 80 bytes
    double mu, Nv, best;
    double pose[3];
    int32_t parent, child0;
    int32_t mask;    // entry of the tree's mask pool, -1 until the node is selected for the first time
    uint8_t nchild, stage, has_roll, seq_len;
    uint8_t acts[DM_MAXH];  // own action sequence from the root (stage entries)
    uint8_t seq[DM_MAXH];   // best roll-out through this node: full action sequence from the root (seq_len entries)
};
struct DmMasks {  // 960 bytes, one per selected node
    unsigned long long observed[IG_BEL];  // cells observed on the way to this node
    unsigned long long best_obs[IG_BEL];  // cells observed along the best roll-out through this node
};

// What a robot has communicated (ActionDistribution, DecMCTS.py:25-60): kept across planning steps.
struct DmPublished {
    int32_t n, seq_len;
    double q[DM_MAXCOMM];
    uint8_t seq0[DM_MAXH];  // action sequence of the best entry
    unsigned long long obs[DM_MAXCOMM][IG_BEL];
};

struct DmParams {
    int R, Ntree, Nsims, horizon, Ncycles, comm_n, node_cap, mask_cap, xdt;
    unsigned int call_base;
    double c_p, gamma, radius, dt, fov, range;
    unsigned long long seed;
};

__device__ __forceinline__ double dm_u01(unsigned long long seed, unsigned int a, unsigned int b) {
    const unsigned long long h = ig_mix64(seed ^ ig_mix64(((unsigned long long)a << 32) | (unsigned long long)b));
    return (double)(h >> 11) * 0x1.0p-53;
}

__device__ __forceinline__ void dm_prim(int k, double& v, double& w) {  // mcts_avail_actions (ig_mcts.py:247-253)
    v = k / 3 == 0 ? 0.0 : (k / 3 == 1 ? 2.0 : 4.0);
    w = k % 3 == 0 ? -0.5 * kPi : (k % 3 == 1 ? 0.0 : 0.5 * kPi);
}

// Tree._expansion (DecMCTS.py:201-231): one child per feasible primitive, in primitive order.  The children get their pose;
// their observed-cells mask is built when (if) they are selected (dm_materialise).  MU: the tree's compact array of node values
// (the top-n scan reads it instead of the 80-byte nodes).
// Round 4: the phase was 13 % of a grow - nine lanes walked the xdt sub-steps of their primitive one dependent distance-field
// gather after the other, then lane 0 wrote up to nine 80-byte nodes field by field.  Now lane (k, sub) of wave 0 evaluates sub-step
// `sub` of primitive k (the headings are a running sum, the positions are accumulated in order from the lanes' displacements: the
// doubles of ig_next_pose, as in dm_next_pose_wave; one gather per lane, all in flight together), a ballot turns the tests into the
// feasible set, and the lane that owns a primitive's last sub-step writes that child.  9 xdt <= 64 (larger: the sequential version).
__device__ inline void dm_expand(const uint32_t* d2, DmNode* T, double* MU, int* n_nodes, int s, const DmParams& P, double* cpose, int* cfeas, int tid) {
    const int stage = T[s].stage;
    if (stage >= P.horizon || T[s].nchild != 0) return;  // uniform: every lane reads the same node
    const int xdt = P.xdt;
    if (9 * xdt <= 64) {
        if (tid < 64) {
            const int lane = tid;
            const bool work = lane < 9 * xdt;
            const int k = work ? lane / xdt : 0, sub = work ? lane - k * xdt : 0, base = k * xdt;
            const double x0 = T[s].pose[0], y0 = T[s].pose[1], th0 = T[s].pose[2];
            double v, w;
            dm_prim(k, v, w);
            double nt = th0;  // heading BEFORE sub-step `sub`
            for (int j = 0; j < xdt; j++)
                if (j < sub) nt = nt + w * P.dt;
            double sn, cs;
            ig_sincos(nt, &sn, &cs);
            const double dx = fma(cs, v, -sn * 0.0) * P.dt, dy = fma(sn, v, cs * 0.0) * P.dt;  // this sub-step's displacement
            double nx = x0, ny = y0;  // position AFTER sub-step `sub`
            for (int j = 0; j < xdt; j++) {
                const double ax = __shfl(dx, base + j, 64), ay = __shfl(dy, base + j, 64);
                if (j <= sub) { nx = nx + ax; ny = ny + ay; }
            }
            bool bad = false;
            if (work && v != 0.0) {
                const bool in_map = (IG_HALF > nx) && (IG_HALF > ny) && (nx > -IG_HALF) && (ny > -IG_HALF);
                bad = !in_map || !(edf_at(d2, nx, ny) > P.radius + 0.1);
            }
            const unsigned long long bm = __ballot(bad);
            const unsigned long long gmask = ((1ull << xdt) - 1ull) << base;
            const bool owner = work && sub == xdt - 1;  // holds the primitive's final position
            const bool feas = owner && (bm & gmask) == 0ull;
            const unsigned long long fm = __ballot(feas);
            const int first = *n_nodes;
            const int room = P.node_cap - first;  // a full pool stops growing (sized so it never is)
            const int idx = __popcll(fm & ((1ull << lane) - 1ull));
            int cnt = __popcll(fm);
            cnt = cnt < room ? cnt : (room > 0 ? room : 0);
            if (feas && idx < cnt) {
                double tt = th0;
                for (int j = 0; j < xdt; j++) tt = tt + w * P.dt;
                DmNode& c = T[first + idx];
                c.mu = 0.0; c.Nv = 0.0; c.best = 0.0;
                c.pose[0] = nx; c.pose[1] = ny; c.pose[2] = tt;
                c.parent = s; c.child0 = -1; c.mask = -1;
                c.nchild = 0; c.stage = (uint8_t)(stage + 1); c.has_roll = 0; c.seq_len = 0;
                for (int a = 0; a < DM_MAXH; a++) { c.acts[a] = a < stage ? T[s].acts[a] : DM_NOACT; c.seq[a] = DM_NOACT; }
                c.acts[stage] = (uint8_t)k;
                MU[first + idx] = 0.0;
            }
            if (lane == 0) {
                T[s].child0 = cnt ? first : -1;
                T[s].nchild = (uint8_t)cnt;
                *n_nodes = first + cnt;
            }
        }
        __syncthreads();
        return;
    }
    if (tid < 9) {
        double x = T[s].pose[0], y = T[s].pose[1], th = T[s].pose[2], v, w;
        dm_prim(tid, v, w);
        const bool ok = ig_next_pose(d2, x, y, th, v, w, P.xdt, P.dt, P.radius);
        cfeas[tid] = ok ? 1 : 0;
        cpose[3 * tid] = x; cpose[3 * tid + 1] = y; cpose[3 * tid + 2] = th;
    }
    __syncthreads();
    if (tid == 0) {
        const int first = *n_nodes;
        int cnt = 0;
        for (int k = 0; k < 9; k++) {
            if (!cfeas[k] || first + cnt >= P.node_cap) continue;  // a full pool stops growing (sized so it never is)
            DmNode& c = T[first + cnt];
            c.mu = 0.0; c.Nv = 0.0; c.best = 0.0;
            c.pose[0] = cpose[3 * k]; c.pose[1] = cpose[3 * k + 1]; c.pose[2] = cpose[3 * k + 2];
            c.parent = s; c.child0 = -1; c.mask = -1;
            c.nchild = 0; c.stage = (uint8_t)(stage + 1); c.has_roll = 0; c.seq_len = 0;
            for (int a = 0; a < DM_MAXH; a++) { c.acts[a] = a < stage ? T[s].acts[a] : DM_NOACT; c.seq[a] = DM_NOACT; }
            c.acts[stage] = (uint8_t)k;
            MU[first + cnt] = 0.0;
            cnt++;
        }
        T[s].child0 = cnt ? first : -1;
        T[s].nchild = (uint8_t)cnt;
        *n_nodes = first + cnt;
    }
    __syncthreads();
}

// The masks of a node that is selected for the first time: observed = parent's observed | cells visible from its pose
// (mcts_sim_state_storer, ig_mcts.py:210-232).  The parent was selected before (it has children), the root at start-up.
__device__ inline void dm_materialise(const uint32_t* d2, DmNode* T, DmMasks* MK, int* n_masks, int s, const DmParams& P,
                                      unsigned long long* vis, int tid, const IgCone& cone) {
    if (T[s].mask >= 0) return;  // uniform
    const int m = *n_masks;      // (read by every lane before lane 0 bumps it below, behind the barriers of the visibility block)
    ig_visible_block(d2, T[s].pose[0], T[s].pose[1], T[s].pose[2], P.fov, P.range, vis, tid, DM_THREADS, &cone);
    const DmMasks& pm = MK[T[T[s].parent].mask];
    for (int j = tid; j < IG_BEL; j += DM_THREADS) {
        MK[m].observed[j] = pm.observed[j] | vis[j];
        MK[m].best_obs[j] = 0ull;
    }
    __syncthreads();
    if (tid == 0) { T[s].mask = m; *n_masks = m + 1; }
    __syncthreads();
}

// ---- one roll-out on ONE wave ---------------------------------------------------------------------------------------------
// targetMap.getVisibleCells on the 64 lanes of a wave into the wave's own LDS mask (no workgroup barrier: a wave's LDS
// operations complete in program order; the fences keep the compiler from reordering them).  Same set as ig_visible_block.
__device__ __forceinline__ void dm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Two passes: (1) every lane classifies its share of the window's cells by the cone test and the cells inside are appended
// to the wave's list (ballot + prefix count); (2) the sphere traces - whose length varies from cell to cell and is what costs -
// run on dense lanes, one listed cell per lane and round (before: a lane traced the 0..3 cone cells that happened to be among
// its own window cells, the wave waiting for the unluckiest lane).
__device__ inline void dm_visible_wave(const uint32_t* d2, double px, double py, double phi, double fov, double range,
                                       unsigned long long* vis, uint16_t* list, int lane, const IgCone& cone) {
    if (lane < IG_BEL) vis[lane] = 0ull;
    dm_wave_sync();
    double s, c;
    ig_sincos(phi, &s, &c);
    double sl, cl, sr, cr;
    ig_sincos(phi + fov, &sl, &cl);
    ig_sincos(phi - fov, &sr, &cr);
    int cx[4] = {ig_bel_cell(px), ig_bel_cell(ig_clamp(px + range * c)), ig_bel_cell(ig_clamp(px + range * cl)),
                 ig_bel_cell(ig_clamp(px + range * cr))};
    int cy[4] = {ig_bel_cell(py), ig_bel_cell(ig_clamp(py + range * s)), ig_bel_cell(ig_clamp(py + range * sl)),
                 ig_bel_cell(ig_clamp(py + range * sr))};
    int xs = min(min(cx[0], cx[1]), min(cx[2], cx[3])), xe = max(max(cx[0], cx[1]), max(cx[2], cx[3]));
    int ys = min(min(cy[0], cy[1]), min(cy[2], cy[3])), ye = max(max(cy[0], cy[1]), max(cy[2], cy[3]));
    xs = max(xs, 0);
    ys = max(ys, 0);
    xe = min(xe, IG_BEL);
    ye = min(ye, IG_BEL);
    const int w = xe - xs, h = ye - ys;
    const int total = (w > 0 && h > 0) ? w * h : 0;
    const bool fast = cone.fast;  // the cone test by products where the margin allows (see ig_visible_block); the tangents: once per kernel
    const double t_in = cone.t_in, t_out = cone.t_out;
    const double r2_in = range * range * (1.0 - 1e-12), r2_out = range * range * (1.0 + 1e-12);
    constexpr int CAP = IG_BEL * IG_BEL / 4;  // list entries of a wave; a larger window is worked off in slices of CAP cells
    for (int base = 0; base < total; base += CAP) {
        const int lim = total - base < CAP ? total : base + CAP;
        int n = 0;  // cells in the cone so far (wave-uniform)
        for (int q0 = base; q0 < lim; q0 += 64) {
            const int q = q0 + lane;
            bool inside = false;
            int i = 0, j = 0;
            if (q < lim) {
                i = xs + q / h;
                j = ys + q % h;
                double cxp = (i)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2, cyp = (j)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2;
                double r0, r1;
                mat2vec(c, s, cxp - px, cyp - py, r0, r1);
                const double rn2 = r0 * r0 + r1 * r1, a1 = fabs(r1);
                if (fast && (rn2 > r2_out || !(r0 > 0.0) || a1 > r0 * t_out)) inside = false;
                else if (fast && rn2 < r2_in && a1 < r0 * t_in) inside = true;
                else {
                    double dphi = atan2(r1, r0);
                    double rn = sqrt(rn2);
                    inside = rn < range && fabs(dphi) < fov / 2;
                }
            }
            const unsigned long long m = __ballot(inside);
            if (inside) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((i << 8) | j);  // i, j < 60
            n += __popcll(m);
        }
        dm_wave_sync();
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            if (e < n) {
                const int i = list[e] >> 8, j = list[e] & 255;
                double cxp = (i)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2, cyp = (j)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2;
                if (ig_check_visibility(d2, px, py, cxp, cyp)) atomicOr(&vis[j], 1ull << i);
            }
        }
        dm_wave_sync();
    }
}

// ig_mcts.get_next_pose (ig_mcts.py:154-183) on one wave: the xdt sub-steps' headings are a running sum (cheap, sequential),
// their sines / cosines and the distance-field tests are independent: lane k < xdt evaluates sub-step k, the positions are then
// accumulated in order from the lanes' velocities (same additions in the same order as ig_next_pose: same doubles).  Uniform
// result on every lane.  xdt <= 64 (larger: the sequential version).
__device__ inline bool dm_next_pose_wave(const uint32_t* d2, double& x, double& y, double& th, double v, double w, int xdt,
                                         double dt, double radius, int lane) {
    if (xdt > 64) return ig_next_pose(d2, x, y, th, v, w, xdt, dt, radius);
    if (v == 0.0) {  // (uniform) a turn on the spot: ig_next_pose adds xdt displacements of exactly +-0.0 and tests nothing
        double tt = th;
        for (int k = 0; k < xdt; k++) tt = tt + w * dt;
        th = tt;
        return true;
    }
    double nt = th;  // heading BEFORE sub-step `lane`: th + w dt + w dt ... (lane additions, in order)
    for (int k = 0; k < xdt; k++)
        if (k < lane) nt = nt + w * dt;
    double sn, cs;
    ig_sincos(nt, &sn, &cs);
    const double vx = fma(cs, v, -sn * 0.0) * dt, vy = fma(sn, v, cs * 0.0) * dt;  // this sub-step's displacement
    double nx = x, ny = y, mx = 0.0, my = 0.0;  // (mx, my): the position AFTER sub-step `lane`
    for (int k = 0; k < xdt; k++) {
        nx = nx + __shfl(vx, k, 64);
        ny = ny + __shfl(vy, k, 64);
        if (k == lane) { mx = nx; my = ny; }
    }
    bool bad = false;
    if (lane < xdt && v != 0.0) {
        const bool in_map = (IG_HALF > mx) && (IG_HALF > my) && (mx > -IG_HALF) && (my > -IG_HALF);
        bad = !in_map || !(edf_at(d2, mx, my) > radius + 0.1);
    }
    if (__ballot(bad) != 0ull) return false;
    double tt = th;
    for (int k = 0; k < xdt; k++) tt = tt + w * dt;
    x = nx;
    y = ny;
    th = tt;
    return true;
}
// MI sum over the cells of `mask` on one wave in the summation order of ig_reward_block with DM_VT = 128 threads: lane l
// carries the partial sums of the virtual threads l and l + 64 (each over q = vt, vt + 128, ... ascending), adds them as
// the reduction's first level does (red[l] += red[l + 64]) and the remaining levels run through lane shuffles.
// Round 4: walked ROW by row.  A virtual thread's cells are q = vt, vt + 128, ...: with 60 cells per row and 64 lanes, lane l meets
// at most ONE cell of row j - column i = (l + 4 j) mod 64, when that is below 60 - and it belongs to virtual thread l or l + 64 by bit 6 of
// q - l.  Rows in ascending order are therefore each virtual thread's cells in ascending q, the order of the block reduction, and a row
// whose mask word is zero (most of them: a roll-out sees a few cones) is skipped for the whole wave.  (The sum was the largest single
// item of a roll-out: 57 iterations with an integer division and a 64-bit shift each, whatever the mask held.)
__device__ inline double dm_reward_wave(const double* belief, const unsigned long long* mask, int lane) {
    double a0 = 0.0, a1 = 0.0;
    // six rows' cached values are requested together (a lane's cell of a row outside the mask requests nothing): the world's MI cache
    // misses the L1 more often than not (eight worlds' distance-field gathers share it), and one request per row was one dependent round
    // trip to the L2 per non-empty row
    constexpr int RB = 6;
#pragma unroll 1
    for (int j0 = 0; j0 < IG_BEL; j0 += RB) {
        double v[RB];
        bool hi[RB], on[RB];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const int j = j0 + u;
            const unsigned long long row = mask[j];  // the same word on every lane
            const int i = (lane + 4 * j) & 63;
            const int q = j * IG_BEL + i;
            on[u] = i < IG_BEL && ((row >> i) & 1ull);
            hi[u] = ((q - lane) >> 6) & 1;
            v[u] = 0.0;
            if (on[u]) v[u] = belief[q];
        }
#pragma unroll
        for (int u = 0; u < RB; u++) {  // rows in ascending order: each virtual thread's cells in ascending q
            if (on[u]) {
                if (hi[u]) a1 += v[u];
                else a0 += v[u];
            }
        }
    }
    double r = a0 + a1;
    for (int s = 32; s > 0; s >>= 1) r = r + __shfl_down(r, s, 64);
    return __shfl(r, 0, 64);
}

// Lanes per world (round 4).  A world's planning step is ONE serial chain of R x Ncycles x Ntree grows, each a chain of dependent
// gathers through the L2-resident distance field (sphere traces): the kernel is latency-bound (VALU busy 44 %), a CU holds 16
// waves at 128 VGPRs, and cfg5 gives a CU 8 worlds.  With 4 waves per world only 4 worlds ran at a time - two rounds of
// workgroups - and the 10 roll-outs of a grow took 3 rounds of waves, the third half empty; with 2 waves per world all 8 run
// at once and the roll-outs take exactly 5 rounds: one round of workgroups of ~1.5x the duration instead of two.
__global__ void __launch_bounds__(DM_THREADS, DM_WAVES_PER_SIMD) k_dmcts_plan(IgDev G, DmParams P, const double* poses, DmNode* nodes, DmMasks* masks, double* mu_all,
                                                         int32_t* n_nodes_all, DmPublished* pub_all, double* out_actions,
                                                         uint8_t* out_paths, double* out_stats) {
    __shared__ unsigned long long vis[IG_BEL], excl[IG_BEL], bobs[IG_BEL];
    __shared__ unsigned long long wvis[DM_WAVES][IG_BEL], wobs[DM_WAVES][IG_BEL], wbobs[DM_WAVES][IG_BEL];  // a wave's own masks
    __shared__ uint16_t wlist[DM_WAVES][IG_BEL * IG_BEL / 4];  // cone cells of the wave's current visibility query (a 60-degree cone of 5 m holds < 100)
    __shared__ uint8_t wtail[DM_WAVES][DM_MAXH], wbtail[DM_WAVES][DM_MAXH];
    __shared__ double wbest[DM_WAVES];
    __shared__ int wbest_sim[DM_WAVES];
    __shared__ double red[DM_THREADS];
    __shared__ int redi[DM_THREADS];
    __shared__ double cpose[27];
    __shared__ int cfeas[9];
    __shared__ double rew[DM_MAXSIMS];
    __shared__ uint8_t btail[DM_MAXH];
    __shared__ int sh_sel, sh_pick, sh_n, sh_depth;
    __shared__ int sh_path[DM_MAXH + 2], sh_better[DM_MAXH + 2];
    __shared__ int picks[DM_MAXCOMM];
    __shared__ int dist_idx[DM_MAXR][DM_MAXCOMM];
    __shared__ double dist_q[DM_MAXR][DM_MAXCOMM];
    __shared__ int dist_n[DM_MAXR];
    const int w = blockIdx.x, tid = threadIdx.x, R = P.R, H = P.horizon;
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t* d2 = G.d2 + (size_t)ig_scenario(G, w) * CAGYM_MAPD * CAGYM_MAPD;
    const double* belief = G.mi + (size_t)w * IG_BEL * IG_BEL;  // the MI cache of the world's belief (IgDev::mi)
    DmPublished* pub = pub_all + (size_t)w * R;
    const IgCone cone = ig_cone(P.fov);  // the cone test's two tangents, once for every visibility query of the planning step

    // ---- Tree.__init__ (DecMCTS.py:92-138): root + expansion of the root; my_act_dist = the root state alone -----
    for (int r = 0; r < R; r++) {
        DmNode* T = nodes + ((size_t)w * R + r) * P.node_cap;
        DmMasks* MK = masks + ((size_t)w * R + r) * P.mask_cap;
        double* MU = mu_all + ((size_t)w * R + r) * P.node_cap;  // the nodes' values once more, 8 bytes apart (the top-n scan's input)
        int* nn = n_nodes_all + ((size_t)w * R + r) * 2;  // [0] nodes, [1] mask-pool entries
        for (int j = tid; j < IG_BEL; j += DM_THREADS) { MK[0].observed[j] = 0ull; MK[0].best_obs[j] = 0ull; }
        if (tid == 0) {
            T[0].mu = 0.0; T[0].Nv = 0.0; T[0].best = 0.0;
            T[0].pose[0] = poses[((size_t)w * R + r) * 3]; T[0].pose[1] = poses[((size_t)w * R + r) * 3 + 1];
            T[0].pose[2] = poses[((size_t)w * R + r) * 3 + 2];
            T[0].parent = -1; T[0].child0 = -1; T[0].mask = 0; T[0].nchild = 0; T[0].stage = 0; T[0].has_roll = 0; T[0].seq_len = 0;
            for (int a = 0; a < DM_MAXH; a++) { T[0].acts[a] = DM_NOACT; T[0].seq[a] = DM_NOACT; }
            MU[0] = 0.0;
            nn[0] = 1; nn[1] = 1;
            dist_n[r] = 1; dist_idx[r][0] = 0; dist_q[r][0] = 1.0;
        }
        __syncthreads();
        dm_expand(d2, T, MU, nn, 0, P, cpose, cfeas, tid);
    }

    for (int cycle = 0; cycle < P.Ncycles; cycle++) {
        for (int r = 0; r < R; r++) {
            DmNode* T = nodes + ((size_t)w * R + r) * P.node_cap;
            DmMasks* MK = masks + ((size_t)w * R + r) * P.mask_cap;
            double* MU = mu_all + ((size_t)w * R + r) * P.node_cap;
            int* nn = n_nodes_all + ((size_t)w * R + r) * 2;
            for (int g = 0; g < P.Ntree; g++) {
                const unsigned int call = P.call_base + (unsigned int)((cycle * R + r) * P.Ntree + g) + 1u;
                DMSTAMP_BEGIN();
                DMSTAMP_COUNT();
                // ---- _get_system_state (DecMCTS.py:182-194): one sampled plan per robot this one listens to -------
                for (int j = tid; j < IG_BEL; j += DM_THREADS) excl[j] = 0ull;
                __syncthreads();
                for (int other = 0; other < R - 1; other++) {  // Q15 listening graph (see dmcts.py)
                    if (other == r || pub[other].n <= 0) continue;  // uniform
                    if (tid == 0) {
                        const int n = pub[other].n;
                        double tot = 0.0;
                        for (int k = 0; k < n; k++) tot += pub[other].q[k];
                        const double thr = dm_u01(P.seed ^ 0x5DEECE66Dull, (unsigned int)w, (call << 4) | (unsigned int)other) * tot;
                        int pick = n - 1;
                        double run = 0.0;
                        for (int k = 0; k < n; k++) {
                            run += pub[other].q[k];
                            if (run > thr) { pick = k; break; }
                        }
                        sh_pick = pick;
                    }
                    __syncthreads();
                    for (int j = tid; j < IG_BEL; j += DM_THREADS) excl[j] |= pub[other].obs[sh_pick][j];
                    __syncthreads();
                }
                DMSTAMP(0);
                // ---- selection (DecMCTS.py:14-18, 140-153, 288-289) --------------------------------------------------
                // (the children of a node are scored side by side on the lanes of wave 0 - a logarithm, a division and a square root
                //  each; lane 0 alone walked up to 4 levels x 9 children - and the first maximum in child order is taken, as `if u > best`)
                if (wave == 0) {
                    int node = 0, depth = 0;
                    for (;;) {
                        if (lane == 0) sh_path[depth] = node;  // the walk from the root: back-propagation climbs it without reloading parents
                        const int nc = T[node].nchild;  // uniform
                        if (nc <= 0) break;
                        depth++;
                        const double n_p = T[node].Nv;
                        const int c0 = T[node].child0;
                        double u = -INFINITY;
                        int bi = lane;
                        if (lane < nc) {
                            const DmNode& ch = T[c0 + lane];
                            if (ch.Nv == 0.0) u = INFINITY;
                            else u = n_p > 0.0 ? ch.mu + 2 * P.c_p * sqrt(2 * log(n_p) / ch.Nv) : ch.mu;
                        }
                        for (int off = 8; off > 0; off >>= 1) {  // nc <= 9: lanes 0..15
                            const double u2 = __shfl_down(u, off, 64);
                            const int i2 = __shfl_down(bi, off, 64);
                            if (u2 > u || (u2 == u && i2 < bi)) { u = u2; bi = i2; }
                        }
                        node = c0 + __shfl(bi, 0, 64);
                    }
                    if (lane == 0) { sh_sel = node; sh_depth = depth; }
                }
                __syncthreads();
                const int s = sh_sel;
                DMSTAMP(1);
                dm_materialise(d2, T, MK, nn + 1, s, P, vis, tid, cone);
                DMSTAMP(2);
                dm_expand(d2, T, MU, nn, s, P, cpose, cfeas, tid);
                DMSTAMP(3);
                // ---- simulation (DecMCTS.py:233-271, 296-327): Nsims random roll-outs from the selected node, one wave each ----
                const int steps = H - T[s].stage;
                const unsigned long long rseed = P.seed * 1000003ull + (unsigned long long)call;
                const DmMasks& sm = MK[T[s].mask];
                {
                    double my_best = -INFINITY;
                    int my_best_sim = 0x7fffffff;
                    for (int sim = wave; sim < P.Nsims; sim += DM_WAVES) {
                        unsigned long long* obs = wobs[wave];
                        if (lane < IG_BEL) obs[lane] = sm.observed[lane];
                        dm_wave_sync();
                        double x = T[s].pose[0], y = T[s].pose[1], th = T[s].pose[2];
                        for (int t = 0; t < steps; t++) {
                            const uint32_t k = ig_rand_primitive(rseed, (uint32_t)w, (uint32_t)sim, (uint32_t)t);
                            double v, wv;
                            dm_prim((int)k, v, wv);
                            const bool ok = dm_next_pose_wave(d2, x, y, th, v, wv, P.xdt, P.dt, P.radius, lane);  // uniform across the wave
                            DMSTAMP(8);
                            if (ok) {
                                dm_visible_wave(d2, x, y, th, P.fov, P.range, wvis[wave], wlist[wave], lane, cone);
                                if (lane < IG_BEL) obs[lane] |= wvis[wave][lane];
                                dm_wave_sync();
                            }
                            if (lane == 0) wtail[wave][t] = ok ? (uint8_t)k : (uint8_t)DM_INFEAS;
                            DMSTAMP(9);
                        }
                        if (lane < IG_BEL) wvis[wave][lane] = obs[lane] & ~excl[lane];  // mcts_reward (ig_mcts.py:234-241)
                        dm_wave_sync();
                        const double rr = dm_reward_wave(belief, wvis[wave], lane);
                        DMSTAMP(10);
                        if (lane == 0) rew[sim] = rr;
                        if (rr > my_best) {  // uniform across the wave; `if rew > best_reward` keeps the first maximum
                            my_best = rr;
                            my_best_sim = sim;
                            if (lane < IG_BEL) wbobs[wave][lane] = obs[lane];
                            if (lane < DM_MAXH) wbtail[wave][lane] = lane < steps ? wtail[wave][lane] : (uint8_t)DM_NOACT;
                        }
                        dm_wave_sync();
                    }
                    if (lane == 0) { wbest[wave] = my_best; wbest_sim[wave] = my_best_sim; }
                }
                DMSTAMP(4);
                __syncthreads();
                DMSTAMP(5);
                // the first maximum in roll-out order: highest reward, lowest roll-out index among equals
                int bw = 0;
                for (int q = 1; q < DM_WAVES; q++)
                    if (wbest[q] > wbest[bw] || (wbest[q] == wbest[bw] && wbest_sim[q] < wbest_sim[bw])) bw = q;
                const double best = wbest[bw];
                for (int j = tid; j < IG_BEL; j += DM_THREADS) bobs[j] = wbobs[bw][j];
                if (tid < DM_MAXH) btail[tid] = wbtail[bw][tid];
                __syncthreads();
                // ---- back-propagation (DecMCTS.py:329-356) ----------------------------------------------------------------
                for (int j = tid; j < IG_BEL; j += DM_THREADS) MK[T[s].mask].best_obs[j] = bobs[j];
                if (tid == 0) {
                    double acc = 0.0;
                    for (int sim = 0; sim < P.Nsims; sim++) acc += rew[sim];
                    const double avg = acc / P.Nsims;
                    const int st = T[s].stage;
                    T[s].mu = avg; T[s].best = best; T[s].Nv = 1.0; T[s].has_roll = 1;
                    MU[s] = avg;
                    T[s].seq_len = (uint8_t)H;
                    for (int a = 0; a < DM_MAXH; a++) T[s].seq[a] = a < st ? T[s].acts[a] : (a < H ? btail[a - st] : (uint8_t)DM_NOACT);
                    red[0] = avg;
                }
                __syncthreads();
                const double avg = red[0];
                // the selected node's ancestors (the walk T[a].parent, recorded at selection).  Each one's update reads and writes only its
                // own statistics, so they are independent: thread d takes ancestor d (round 4; lane 0 climbed them one after the other with
                // two barriers and a dependent round trip to the node pool per level)
                if (tid < sh_depth) {
                    const int a = sh_path[tid];
                    const bool better = best > T[a].best;
                    sh_better[tid] = better ? T[a].mask : -1;
                    const double mu_new = (P.gamma * T[a].mu * T[a].Nv + avg) / (T[a].Nv + 1);
                    T[a].mu = mu_new;
                    MU[a] = mu_new;
                    T[a].Nv = P.gamma * T[a].Nv + 1;
                    if (better) {
                        T[a].best = best;
                        T[a].has_roll = 1;
                        T[a].seq_len = T[s].seq_len;
                        for (int q = 0; q < DM_MAXH; q++) T[a].seq[q] = T[s].seq[q];
                    }
                }
                __syncthreads();
                for (int d = 0; d < sh_depth; d++) {
                    const int m = sh_better[d];  // uniform
                    if (m >= 0)
                        for (int j = tid; j < IG_BEL; j += DM_THREADS) MK[m].best_obs[j] = bobs[j];
                }
                DMSTAMP(6);
                // ---- _update_distribution (DecMCTS.py:162-180): top comm_n nodes by mu (first created first on ties),
                //      those with a roll-out, q = mu^2 -------------------------------------------------------------------------
                const int total = nn[0];
                int mypick[DM_MAXCOMM];  // the rounds' winners (uniform: every thread derives them from the waves' candidates)
                // Thread t owns the nodes 1 + t + k DM_THREADS.  Its best one is found ONCE (round 4: from the tree's compact value array -
                // 8-byte stride, coalesced - instead of the 80-byte nodes); a round's winner is taken out of the race by its owner alone,
                // which looks for its next best - every other thread's candidate stands.  (Before: every thread rescanned all its nodes
                // against the list of winners in each of the comm_n rounds: 11 - 12 % of a grow once a tree holds ~1000 nodes.)
                double bm = -INFINITY;
                int bi = 0x7fffffff;
                for (int i = 1 + tid; i < total; i += DM_THREADS) {  // ascending i: the first maximum wins
                    const double m = MU[i];
                    if (m > bm) { bm = m; bi = i; }
                }
#pragma unroll
                for (int round = 0; round < DM_MAXCOMM; round++) {  // (unrolled to its compile-time bound: mypick stays in registers)
                    if (round >= P.comm_n) break;
                    // arg-max over (mu desc, index asc): inside the wave by lane shuffles, across the waves through LDS - one barrier per round
                    double rm = bm;
                    int ri = bi;
                    for (int off = 32; off > 0; off >>= 1) {
                        const double m2 = __shfl_down(rm, off, 64);
                        const int i2 = __shfl_down(ri, off, 64);
                        if (m2 > rm || (m2 == rm && i2 < ri)) { rm = m2; ri = i2; }
                    }
                    double* wr = red + (round & 1) * DM_WAVES;
                    int* wi = redi + (round & 1) * DM_WAVES;
                    if (lane == 0) { wr[wave] = rm; wi[wave] = ri; }
                    __syncthreads();
                    double gm = wr[0];
                    int gi = wi[0];
                    for (int q = 1; q < DM_WAVES; q++)
                        if (wr[q] > gm || (wr[q] == gm && wi[q] < gi)) { gm = wr[q]; gi = wi[q]; }
                    mypick[round] = gi;
                    if (tid == 0) picks[round] = gi;
                    if (gi == bi && gi != 0x7fffffff) {  // my node won: my next best among the ones not picked yet
                        bm = -INFINITY;
                        bi = 0x7fffffff;
                        for (int i = 1 + tid; i < total; i += DM_THREADS) {
                            bool taken = false;
#pragma unroll
                            for (int q = 0; q < DM_MAXCOMM; q++)
                                if (q <= round) taken |= mypick[q] == i;
                            const double m = MU[i];
                            if (!taken && m > bm) { bm = m; bi = i; }
                        }
                    }
                }
                __syncthreads();
                // those of the winners that have a roll-out, in rank order, with q = mu^2 / sum: lane q of wave 0 fetches winner q's fields
                // (one round trip for all of them; lane 0 alone made up to fifteen dependent ones), the sum runs over the kept entries in order
                if (wave == 0) {
                    const int pk = lane < P.comm_n ? picks[lane] : 0x7fffffff;
                    const bool ok = pk != 0x7fffffff && T[pk].has_roll;
                    const double mu = ok ? T[pk].mu : 0.0;
                    const unsigned long long km = __ballot(ok);
                    const int cnt = __popcll(km), pos = __popcll(km & ((1ull << lane) - 1ull));
                    const double sq = mu * mu;
                    double tot = 0.0;
                    for (int k = 0; k < DM_MAXCOMM; k++) {
                        const double v = __shfl(sq, k, 64);
                        if ((km >> k) & 1ull) tot += v;
                    }
                    if (ok) {  // (an empty list leaves the distribution as it was)
                        dist_idx[r][pos] = pk;
                        dist_q[r][pos] = tot == 0.0 ? 1.0 / cnt : sq / tot;
                    }
                    if (lane == 0) {
                        sh_n = cnt;
                        if (cnt > 0) dist_n[r] = cnt;
                    }
                }
                __syncthreads();
                DMSTAMP(7);
                DMSTAMP_FLUSH();
            }
            // ---- send_comms: publish this robot's distribution (ig_mcts.py:107) ------------------------------------------
            const int n = dist_n[r];
            for (int q = 0; q < n; q++) {
                const DmNode& src = T[dist_idx[r][q]];
                // the root entry of a tree without roll-outs has no best roll-out: it stands for "stay", nothing observed
                const DmMasks& srm = MK[src.mask];  // (distribution entries have a roll-out or are the root: their masks exist)
                for (int j = tid; j < IG_BEL; j += DM_THREADS) pub[r].obs[q][j] = src.has_roll ? srm.best_obs[j] : srm.observed[j];
            }
            if (tid == 0) {
                pub[r].n = n;
                for (int q = 0; q < n; q++) pub[r].q[q] = dist_q[r][q];
                const DmNode& b = T[dist_idx[r][0]];
                pub[r].seq_len = b.has_roll ? b.seq_len : 0;
                for (int a = 0; a < DM_MAXH; a++) pub[r].seq0[a] = b.has_roll ? b.seq[a] : (uint8_t)DM_NOACT;
            }
            __syncthreads();
        }
    }
    // ---- action = first action of the best path (ig_mcts.py:109) --------------------------------------------------------
    if (tid < R) {
        const int r = tid;
        const DmPublished& p = pub[r];
        double v = 0.0, wv = 0.0;
        if (p.seq_len > 0 && p.seq0[0] < 9) dm_prim(p.seq0[0], v, wv);
        out_actions[((size_t)w * R + r) * 2] = v;
        out_actions[((size_t)w * R + r) * 2 + 1] = wv;
        for (int a = 0; a < DM_MAXH; a++) out_paths[((size_t)w * R + r) * DM_MAXH + a] = a < p.seq_len ? p.seq0[a] : (uint8_t)DM_NOACT;
        const DmNode* T = nodes + ((size_t)w * R + r) * P.node_cap;
        out_stats[((size_t)w * R + r) * 3] = T[0].mu;
        out_stats[((size_t)w * R + r) * 3 + 1] = T[0].Nv;
        out_stats[((size_t)w * R + r) * 3 + 2] = (double)n_nodes_all[((size_t)w * R + r) * 2];
    }
}
