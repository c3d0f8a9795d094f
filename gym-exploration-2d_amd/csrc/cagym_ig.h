// cagym_ig.h -- information-gain kernels (gfx950): Euclidean distance field, sphere-traced visibility,
// Bayesian belief update, mutual-information reward, motion primitives, batched random roll-outs.
//
// Replaces information_models/edfMap.py, information_models/targetMap.py and the planner primitives of
// policies/ig_mcts.py:154-253 + pydecmcts/DecMCTS.py:233-271 (reference paths under
// gym_collision_avoidance/envs/).  The tree itself: cagym_dmcts.h (device) or dmcts.py (host).
//
// Memory: per scenario a 300x300 u32 field of SQUARED cell distances (360 KB, L2/MALL resident;
// EDF = sqrt(d2) * 0.1 exactly as scipy's exact EDT), per world a 60x60 fp64 belief grid of odds ratios.
// A visibility set is a 60 x u64 bit mask (bit i of word j <=> cell (i, j)).  This part of the path is
// latency/L2-bound gather work (the reference spends 66 % of step time here), not HBM streaming.
#pragma once
#include "cagym_device.h"

#define IG_BEL 60
#define IG_HALF 15.0
#define IG_EDF_CELL 0.1
#define IG_BEL_CELL 0.5

struct IgDev {
    int N, S;
    const uint32_t* map_bits;  // [S,300,10]
    const int32_t* sc_nobst;   // [S]
    const int32_t* episode;    // [N]
    uint32_t* d2;              // [S,300,300] squared cell distances (exact EDT).  Round 3 measured a pre-multiplied fp64 field
                               //             (sqrt(d2) * 0.1 stored: one 8-byte load instead of load + conversion + fp64 sqrt + product,
                               //             30 of a trace step's ~135 instructions): 5 % SLOWER (321 -> 304 M visibility queries/s) -
                               //             the traces are bound by the gathers (twice the footprint), not by the arithmetic.
    double* belief;            // [N,60,60]
    double* mi;                // [N,60,60] ig_cell_mi(belief): the mutual information of a cell depends on its belief alone, which only
                               //           k_ig_fill_belief / k_ig_update change - the planner's thousands of reward sums per step read
                               //           this cache instead of evaluating four logarithms per observed cell (same doubles, same order)
};

__device__ __forceinline__ int ig_scenario(const IgDev& G, int world) {
    return (int)(((long long)world + (long long)G.episode[world] * G.N) % G.S);
}

// sin and cos of a pose heading (|x| < 1e5 takes the short path: Cody-Waite reduction by pi/2 in two fma steps, the fdlibm
// kernel polynomials on [-pi/4, pi/4]; < 1 ulp like the library's, a third of its instructions).  The roll-outs evaluate
// eight of these per step (five motion sub-steps, three for the view cone): with the library's sincos they were half of
// the planner's arithmetic (profiles/r3/cfg5_*).  Anything else (huge, NaN) goes to the library.
__device__ __forceinline__ void ig_sincos(double x, double* sn, double* cs) {
    if (!(fabs(x) < 1e5)) { sincos(x, sn, cs); return; }
    const double kf = rint(x * 6.36619772367581382433e-01);  // 2 / pi
    double r = fma(-kf, 1.57079632679489655800e+00, x);      // pi/2 rounded; the product is exact inside the fma
    r = fma(-kf, 6.12323399573676603587e-17, r);             // pi/2 - the above
    const double z = r * r;
    // __kernel_sin / __kernel_cos (fdlibm, public domain constants)
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                              -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                              2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int k = (int)kf & 3;
    const double s0 = (k & 1) ? cr : sr, c0 = (k & 1) ? sr : cr;
    *sn = (k & 2) ? -s0 : s0;
    *cs = ((k + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ void mat2vec(double c, double s, double vx, double vy, double& r0, double& r1) {
    r0 = fma(c, vx, s * vy);  // np.dot(((c, s), (-s, c)), v) as executed by the reference (dgemv)
    r1 = fma(-s, vx, c * vy);
}

// edfMap.get_edf_value_from_pose (edfMap.py:14-19)
__device__ __forceinline__ double edf_at(const uint32_t* d2, double x, double y) {
    // floor((x + 15) / 0.1): the product with 10.0 differs from the correctly rounded quotient by a few ulp, so the floors agree
    // unless the product is within 1e-7 of an integer - then (and for huge / NaN operands) the division decides.  The two
    // divisions were 50 of the ~135 instructions of a trace step, and the traces are arithmetic-bound.
    const double tx = x + IG_HALF, ty = y + IG_HALF;
    double qx = tx * 10.0, qy = ty * 10.0;
    if (!(fabs(qx - rint(qx)) > 1e-7 && fabs(qy - rint(qy)) > 1e-7 && fabs(qx) < 1e6 && fabs(qy) < 1e6)) {
        qx = tx / IG_EDF_CELL;
        qy = ty / IG_EDF_CELL;
    }
    double fx = floor(qx), fy = floor(qy);
    if (!(fx > -1e6 && fx < 1e6 && fy > -1e6 && fy < 1e6)) return 0.0;
    int xi = (int)fx, yi = (int)fy;
    if (xi < 0) xi += CAGYM_MAPD;
    if (yi < 0) yi += CAGYM_MAPD;
    if (xi < 0 || yi < 0 || xi >= CAGYM_MAPD || yi >= CAGYM_MAPD) return 0.0;
    return sqrt((double)d2[yi * CAGYM_MAPD + xi]) * IG_EDF_CELL;
}

// edfMap.checkVisibility (edfMap.py:21-47); the trip count is bounded (>= 1e-3 / dist progress per trip)
__device__ inline bool ig_check_visibility(const uint32_t* d2, double px, double py, double gx, double gy) {
    double dx = gx - px, dy = gy - py;
    double dist = sqrt(dx * dx + dy * dy);
    double u = 0.05 / dist;
    for (int guard = 0; u < 1 && guard < 100000; guard++) {
        double nx = (1 - u) * px + u * gx, ny = (1 - u) * py + u * gy;
        double md = edf_at(d2, nx, ny);
        if (md < 0.001) return false;
        u += md / dist;
    }
    return !(u < 1);
}

__device__ __forceinline__ int ig_bel_cell(double v) {
    double f = floor((v + IG_HALF) * 2.0);  // / 0.5: a power of two, the product is the quotient bit for bit
    f = f < -1e6 ? -1e6 : (f > 1e6 ? 1e6 : f);
    return (int)f;
}
__device__ __forceinline__ double ig_clamp(double v) {
    double a = v < IG_HALF ? v : IG_HALF;
    return a > -IG_HALF ? a : -IG_HALF;
}

// The two tangents of the cone test's fast path (below): they depend on the field of view alone.  A caller that asks for many
// visibility sets with one field of view (the planner: one per roll-out step) evaluates them once - two fp64 library tangents
// were ~300 of a query's instructions.
struct IgCone {
    double t_in, t_out;
    bool fast;
};
__device__ __forceinline__ IgCone ig_cone(double fov) {
    const double hf = fov / 2;
    IgCone c;
    c.fast = hf > 1e-6 && hf < 1.5;
    c.t_in = c.fast ? tan(hf - 1e-9) : 0.0;
    c.t_out = c.fast ? tan(hf + 1e-9) : 0.0;
    return c;
}

// targetMap.getVisibleCells (targetMap.py:43-84), computed by all `nthreads` threads of a block into the
// LDS mask `vis[60]` (zeroed here).  Ends with a barrier.
__device__ inline void ig_visible_block(const uint32_t* d2, double px, double py, double phi, double fov, double range,
                                        unsigned long long* vis, int tid, int nthreads, const IgCone* cone = nullptr) {
    for (int j = tid; j < IG_BEL; j += nthreads) vis[j] = 0ull;
    __syncthreads();
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
    // The cone test `rn < range and |atan2(r1, r0)| < fov / 2` (targetMap.py:66-70) costs an fp64 atan2 and a square root per
    // window cell, and two cells in three fail it.  A cell that is inside or outside by a margin far above the rounding of
    // either side is decided by products alone (|r1| against r0 tan(fov/2 -+ 1e-9), rn^2 against range^2 (1 -+ 1e-12)); only
    // a cell within those margins evaluates the reference's own expressions.  Same set, bit for bit.
    const IgCone cn = cone ? *cone : ig_cone(fov);
    const bool fast = cn.fast;
    const double t_in = cn.t_in, t_out = cn.t_out;
    const double r2_in = range * range * (1.0 - 1e-12), r2_out = range * range * (1.0 + 1e-12);
    for (int q = tid; q < total; q += nthreads) {
        int i = xs + q / h, j = ys + q % h;
        double cxp = (i)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2, cyp = (j)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2;
        double r0, r1;
        mat2vec(c, s, cxp - px, cyp - py, r0, r1);
        const double rn2 = r0 * r0 + r1 * r1, a1 = fabs(r1);
        bool inside;
        if (fast && (rn2 > r2_out || !(r0 > 0.0) || a1 > r0 * t_out)) inside = false;
        else if (fast && rn2 < r2_in && a1 < r0 * t_in) inside = true;
        else {
            double dphi = atan2(r1, r0);
            double rn = sqrt(rn2);
            inside = rn < range && fabs(dphi) < fov / 2;
        }
        if (inside) {
            if (ig_check_visibility(d2, px, py, cxp, cyp)) atomicOr(&vis[j], 1ull << i);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double ig_cell_mi(double r) {
    const double rO = 1.5, rE = 0.66, fn = 0.1, fp = 0.05;
    double p = r / (r + 1);
    double f_p = log((r + 1) / (r + (1 / rO))) - log(rO) / (r * rO + 1);
    double f_n = log((r + 1) / (r + (1 / rE))) - log(rE) / (r * rE + 1);
    double P_p = p * (1 - fn) + (1 - p) * fp;
    double P_n = p * fn + (1 - p) * (1 - fp);
    return P_p * f_p + P_n * f_n;
}

// block-wide MI sum over the cells of `mask` (LDS), deterministic order.  red: LDS [nthreads] doubles.
// mi: the world's cache of ig_cell_mi(belief) (IgDev::mi)
__device__ inline double ig_reward_block(const double* mi, const unsigned long long* mask, double* red, int tid,
                                         int nthreads) {
    double acc = 0.0;
    for (int q = tid; q < IG_BEL * IG_BEL; q += nthreads) {
        int j = q / IG_BEL, i = q - j * IG_BEL;
        if ((mask[j] >> i) & 1ull) acc += mi[q];
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = nthreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}

// ig_mcts.get_next_pose (ig_mcts.py:154-183)
__device__ inline bool ig_next_pose(const uint32_t* d2, double& x, double& y, double& th, double v, double w, int xdt,
                                    double dt, double radius) {
    double nx = x, ny = y, nt = th;
    for (int k = 0; k < xdt; k++) {
        double s, c;
        ig_sincos(nt, &s, &c);
        double vx = fma(c, v, -s * 0.0), vy = fma(s, v, c * 0.0);
        nx = nx + vx * dt;
        ny = ny + vy * dt;
        nt = nt + w * dt;
        if (v == 0.0) continue;
        bool in_map = (IG_HALF > nx) && (IG_HALF > ny) && (nx > -IG_HALF) && (ny > -IG_HALF);
        if (!in_map) return false;
        if (!(edf_at(d2, nx, ny) > radius + 0.1)) return false;
    }
    x = nx;
    y = ny;
    th = nt;
    return true;
}

__device__ __forceinline__ uint64_t ig_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t ig_rand_primitive(uint64_t seed, uint32_t q, uint32_t sim, uint32_t step) {
    uint64_t h = ig_mix64(seed ^ ig_mix64(((uint64_t)q << 32) | ((uint64_t)sim << 8) | step));
    return (uint32_t)(h % 9ull);
}

// ---------------------------------------------------------------------------------------------------
// EDT, pass 1: per column, distance (in cells) to the nearest occupied cell of that column.
__global__ void __launch_bounds__(320) k_ig_edt_cols(IgDev G, uint32_t* any_flag) {
    const int s = blockIdx.x, x = threadIdx.x;
    if (x >= CAGYM_MAPD) return;
    const uint32_t* map = G.map_bits + (size_t)s * CAGYM_MAPD * CAGYM_MAPW;
    uint32_t* g = G.d2 + (size_t)s * CAGYM_MAPD * CAGYM_MAPD;
    const int INF = 1 << 20;
    int last = -INF;
    bool any = false;
    for (int y = 0; y < CAGYM_MAPD; y++) {
        if (map_bit(map, y, x)) { last = y; any = true; }
        g[y * CAGYM_MAPD + x] = (uint32_t)(y - last);
    }
    last = INF;
    for (int y = CAGYM_MAPD - 1; y >= 0; y--) {
        if (map_bit(map, y, x)) last = y;
        uint32_t d = (uint32_t)(last - y);
        if (d < g[y * CAGYM_MAPD + x]) g[y * CAGYM_MAPD + x] = d;
    }
    if (any) atomicOr(&any_flag[s], 1u);
}

// EDT, pass 2: per row, lower envelope by brute force (300 candidates per cell), in place.  Every quantity fits 32 bits - a finite
// column distance is <= 299, so a squared distance is <= 2 * 299^2 - and the candidates' squares are taken once per row (round 4: the
// loop ran on 64-bit integers with a multiplication per candidate, 66.6 ms per scenario upload against 6.7 ms for pass 1).
__global__ void __launch_bounds__(320) k_ig_edt_rows(IgDev G, const uint32_t* any_flag) {
    __shared__ uint32_t g2[CAGYM_MAPD];  // squared column distance of the row's cells; 0x7fff0000 = no occupied cell in that column
    const int s = blockIdx.x / CAGYM_MAPD, y = blockIdx.x % CAGYM_MAPD, x = threadIdx.x;
    uint32_t* row = G.d2 + ((size_t)s * CAGYM_MAPD + y) * CAGYM_MAPD;
    if (x < CAGYM_MAPD) {
        const uint32_t gy = row[x];
        g2[x] = gy >= (1u << 19) ? 0x7fff0000u : gy * gy;
    }
    __syncthreads();
    if (x >= CAGYM_MAPD) return;
    uint32_t best = 2u * CAGYM_MAPD * CAGYM_MAPD;
    if (any_flag[s]) {
        for (int xx = 0; xx < CAGYM_MAPD; xx++) {
            const int dx = x - xx;
            const uint32_t d = (uint32_t)(dx * dx) + g2[xx];  // (< 2^31 + 2^17: no wrap; a column without an occupied cell never wins)
            best = d < best ? d : best;
        }
    }
    row[x] = best;
}

__global__ void __launch_bounds__(256) k_ig_fill_belief(IgDev G, const uint8_t* mask) {
    const int w = blockIdx.x;
    if (mask && !mask[w]) return;
    double* b = G.belief + (size_t)w * IG_BEL * IG_BEL;
    double* m = G.mi + (size_t)w * IG_BEL * IG_BEL;
    const double mi1 = ig_cell_mi(1.0);
    for (int q = threadIdx.x; q < IG_BEL * IG_BEL; q += blockDim.x) { b[q] = 1.0; m[q] = mi1; }  // prior (targetMap.py:8)
}

__global__ void __launch_bounds__(128) k_ig_visible(IgDev G, const double* poses, const int32_t* world, double fov,
                                                    double range, unsigned long long* masks) {
    __shared__ unsigned long long vis[IG_BEL];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (world[q] < 0 || world[q] >= G.N) {  // caller-supplied index out of range: empty set, nothing is read
        for (int j = tid; j < IG_BEL; j += blockDim.x) masks[(size_t)q * IG_BEL + j] = 0ull;
        return;
    }
    const uint32_t* d2 = G.d2 + (size_t)ig_scenario(G, world[q]) * CAGYM_MAPD * CAGYM_MAPD;
    ig_visible_block(d2, poses[3 * q], poses[3 * q + 1], poses[3 * q + 2], fov, range, vis, tid, blockDim.x);
    for (int j = tid; j < IG_BEL; j += blockDim.x) masks[(size_t)q * IG_BEL + j] = vis[j];
}

// targetMap.update (targetMap.py:86-128): one block per world, poses applied in order.
__global__ void __launch_bounds__(256) k_ig_update(IgDev G, const double* poses, const int32_t* n_poses,
                                                   const double* dets, const int32_t* n_det, int P, int Dmax,
                                                   double fov, double range, unsigned long long* observed) {
    __shared__ unsigned long long vis[IG_BEL];
    __shared__ unsigned long long uni[IG_BEL];
    const int w = blockIdx.x, tid = threadIdx.x;
    const uint32_t* d2 = G.d2 + (size_t)ig_scenario(G, w) * CAGYM_MAPD * CAGYM_MAPD;
    double* bel = G.belief + (size_t)w * IG_BEL * IG_BEL;
    for (int j = tid; j < IG_BEL; j += blockDim.x) uni[j] = 0ull;
    __syncthreads();
    const int np = n_poses ? n_poses[w] : P;
    const double thr = sqrt(0.5) * IG_BEL_CELL + 0.01;
    for (int p = 0; p < np; p++) {
        const double* pose = poses + ((size_t)w * P + p) * 3;
        const double px = pose[0], py = pose[1], phi = pose[2];
        ig_visible_block(d2, px, py, phi, fov, range, vis, tid, blockDim.x);
        double s, c;
        ig_sincos(phi, &s, &c);
        const int nd = n_det[w * P + p];
        for (int q = tid; q < IG_BEL * IG_BEL; q += blockDim.x) {
            int j = q / IG_BEL, i = q - j * IG_BEL;
            if (!((vis[j] >> i) & 1ull)) continue;
            double rs = 0.66;
            if (nd > 0) {
                double cx = (i)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2, cy = (j)*IG_BEL_CELL - IG_HALF + IG_BEL_CELL / 2;
                double r0, r1;
                mat2vec(c, s, cx - px, cy - py, r0, r1);
                for (int d = 0; d < nd; d++) {
                    const double* tg = dets + (((size_t)w * P + p) * Dmax + d) * 2;
                    double t0, t1;
                    mat2vec(c, s, tg[0] - px, tg[1] - py, t0, t1);
                    double e0 = t0 - r0, e1 = t1 - r1;
                    if (sqrt(e0 * e0 + e1 * e1) < thr) { rs = 1.5; break; }
                }
            }
            bel[q] *= rs;
        }
        for (int j = tid; j < IG_BEL; j += blockDim.x) uni[j] |= vis[j];
        __syncthreads();
    }
    if (observed)
        for (int j = tid; j < IG_BEL; j += blockDim.x) observed[(size_t)w * IG_BEL + j] = uni[j];
    // the MI cache follows the belief (only cells seen in this update changed)
    double* mi = G.mi + (size_t)w * IG_BEL * IG_BEL;
    for (int q = tid; q < IG_BEL * IG_BEL; q += blockDim.x) {
        int j = q / IG_BEL, i = q - j * IG_BEL;
        if ((uni[j] >> i) & 1ull) mi[q] = ig_cell_mi(bel[q]);
    }
}

__global__ void __launch_bounds__(256) k_ig_reward(IgDev G, const unsigned long long* masks, const int32_t* world,
                                                   double* reward) {
    __shared__ unsigned long long m[IG_BEL];
    __shared__ double red[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (world[q] < 0 || world[q] >= G.N) {  // out-of-range world: zero reward
        if (tid == 0) reward[q] = 0.0;
        return;
    }
    for (int j = tid; j < IG_BEL; j += blockDim.x) m[j] = masks[(size_t)q * IG_BEL + j];
    __syncthreads();
    double r = ig_reward_block(G.mi + (size_t)world[q] * IG_BEL * IG_BEL, m, red, tid, blockDim.x);
    if (tid == 0) reward[q] = r;
}

__global__ void __launch_bounds__(256) k_ig_next_pose(IgDev G, const double* poses, const double* actions,
                                                      const int32_t* world, const double* radius, int Q, int xdt,
                                                      double dt, double* next, uint8_t* feasible) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    if (world[q] < 0 || world[q] >= G.N) {  // out-of-range world: infeasible, pose unchanged
        next[3 * q] = poses[3 * q]; next[3 * q + 1] = poses[3 * q + 1]; next[3 * q + 2] = poses[3 * q + 2];
        feasible[q] = 0;
        return;
    }
    const uint32_t* d2 = G.d2 + (size_t)ig_scenario(G, world[q]) * CAGYM_MAPD * CAGYM_MAPD;
    double x = poses[3 * q], y = poses[3 * q + 1], th = poses[3 * q + 2];
    bool ok = ig_next_pose(d2, x, y, th, actions[2 * q], actions[2 * q + 1], xdt, dt, radius[q]);
    next[3 * q] = x;
    next[3 * q + 1] = y;
    next[3 * q + 2] = th;
    feasible[q] = ok ? 1 : 0;
}

// Tree._simulate (DecMCTS.py:233-271) with mcts_sim_state_storer / mcts_reward (ig_mcts.py:210-241):
// one block per (query, sim).  Every thread advances the (tiny) pose recurrence redundantly; the
// visibility query of each step is spread over the block.
__global__ void __launch_bounds__(128) k_ig_rollouts(IgDev G, const double* pose0, const unsigned long long* observed0,
                                                     const unsigned long long* exclude, const int32_t* world,
                                                     const int32_t* n_steps, const double* radius, int nsims,
                                                     int max_steps, int xdt, double dt, double fov, double range,
                                                     unsigned long long seed, double* rewards, uint8_t* actions,
                                                     double* final_pose, unsigned long long* observed_out) {
    __shared__ unsigned long long vis[IG_BEL];
    __shared__ unsigned long long obs[IG_BEL];
    __shared__ double red[128];
    const int q = blockIdx.x / nsims, sim = blockIdx.x % nsims, tid = threadIdx.x;
    const int w = world[q];
    if (w < 0 || w >= G.N) {  // out-of-range world: zero reward, no roll-out
        if (tid == 0) {
            rewards[(size_t)q * nsims + sim] = 0.0;
            if (final_pose) {
                double* fp = final_pose + ((size_t)q * nsims + sim) * 3;
                fp[0] = pose0[3 * q]; fp[1] = pose0[3 * q + 1]; fp[2] = pose0[3 * q + 2];
            }
        }
        if (actions) for (int t = tid; t < max_steps; t += blockDim.x) actions[((size_t)q * nsims + sim) * max_steps + t] = 255;
        if (observed_out) for (int j = tid; j < IG_BEL; j += blockDim.x) observed_out[((size_t)q * nsims + sim) * IG_BEL + j] = 0ull;
        return;
    }
    const uint32_t* d2 = G.d2 + (size_t)ig_scenario(G, w) * CAGYM_MAPD * CAGYM_MAPD;
    for (int j = tid; j < IG_BEL; j += blockDim.x) obs[j] = observed0[(size_t)q * IG_BEL + j];
    __syncthreads();
    double x = pose0[3 * q], y = pose0[3 * q + 1], th = pose0[3 * q + 2];
    int T = n_steps[q];
    if (T > max_steps) T = max_steps;
    const double rad = radius[q];
    for (int t = 0; t < T; t++) {
        uint32_t k = ig_rand_primitive(seed, (uint32_t)q, (uint32_t)sim, (uint32_t)t);
        const double v = k / 3 == 0 ? 0.0 : (k / 3 == 1 ? 2.0 : 4.0);
        const double wv = k % 3 == 0 ? -0.5 * kPi : (k % 3 == 1 ? 0.0 : 0.5 * kPi);
        bool ok = ig_next_pose(d2, x, y, th, v, wv, xdt, dt, rad);  // uniform across the block
        if (ok) {
            ig_visible_block(d2, x, y, th, fov, range, vis, tid, blockDim.x);
            for (int j = tid; j < IG_BEL; j += blockDim.x) obs[j] |= vis[j];
            __syncthreads();
        }
        if (tid == 0 && actions) actions[((size_t)q * nsims + sim) * max_steps + t] = ok ? (uint8_t)k : (uint8_t)255;
    }
    if (observed_out)  // cells observed along this roll-out (incl. observed0), BEFORE other robots' cells are removed
        for (int j = tid; j < IG_BEL; j += blockDim.x) observed_out[((size_t)q * nsims + sim) * IG_BEL + j] = obs[j];
    for (int j = tid; j < IG_BEL; j += blockDim.x) obs[j] &= ~exclude[(size_t)q * IG_BEL + j];
    __syncthreads();
    double r = ig_reward_block(G.mi + (size_t)w * IG_BEL * IG_BEL, obs, red, tid, blockDim.x);
    if (tid == 0) {
        rewards[(size_t)q * nsims + sim] = r;
        if (final_pose) {
            double* fp = final_pose + ((size_t)q * nsims + sim) * 3;
            fp[0] = x; fp[1] = y; fp[2] = th;
        }
    }
}
