/* cagym_oracle.c -- CPU restatement (parity oracle) of the reference env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cagym_oracle.h).  Plain C, scalar, fp64 state exactly as
 * the NumPy reference keeps it (agent.py:21-25); actions pass through fp32
 * (collision_avoidance_env.py:289).  Compile with -ffp-contract=off so no FMA is formed.
 *
 * Citations: `env.py` = gym_collision_avoidance/envs/collision_avoidance_env.py, other
 * paths relative to gym_collision_avoidance/envs/ of the reference.
 */
#include "cagym_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAPD 300 /* int(30/0.1), Map.py:18 */
#define NBEAM 16 /* Config.LASERSCAN_LENGTH, config.py:57 */
#define NSAMP 16 /* len(arange(0, 6, 2*pi/16)), sensors/LaserScanSensor.py:20 */

static const double PI = 3.141592653589793; /* np.pi */

struct cao_env {
    cao_config cfg;
    int N, M, K, Kobs; /* K = M-1 OAS rows */
    /* scenario */
    double *sc_agents6, *sc_heading0, *sc_coop, *sc_obst;
    int32_t *sc_policy, *sc_dyn, *sc_nagents, *sc_nobst;
    uint8_t* sc_has_heading;
    /* state, all [N*M*w] */
    double* f[CAO_F_COUNT];
    uint8_t* u[CAO_U_COUNT];
    int32_t* i32[CAO_I_COUNT];
    double *goal, *radius, *pref_speed, *ref_orth, *ang_speed, *cur_speed, *cur_turn, *coop;
    int32_t *policy, *dyn, *nagents, *nobst;
};

static const int F_WIDTH[CAO_F_COUNT] = {2, 2, 1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 4, 1, 0 /*oas*/, NBEAM, 2};

/* util.py:27-32 */
static double wrap(double a) {
    while (a >= PI) a -= 2 * PI;
    while (a < -PI) a += 2 * PI;
    return a;
}
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* np.dot of 2-vectors and np.linalg.norm (= sqrt(x.dot(x))).  As executed for the golden
 * vectors (NumPy 2.2.6 + OpenBLAS 0.3.29 Haswell ddot) the 2-element dot product is
 * fma(a1, b1, a0*b0): verified bit-for-bit on 20 000 random vectors ("BLAS
 * rounding").  Call sites that use math.sqrt(x**2 + y**2) instead stay plain. */
static double dot2(double a0, double a1, double b0, double b1) { return fma(a1, b1, a0 * b0); }
static double norm2(double x, double y) { return sqrt(dot2(x, y, x, y)); }

cao_env* cao_create(const cao_config* cfg) {
    cao_env* e = (cao_env*)calloc(1, sizeof(cao_env));
    e->cfg = *cfg;
    int N = e->N = cfg->n_worlds, M = e->M = cfg->max_agents;
    e->K = M - 1;
    e->Kobs = cfg->max_obstacles;
    size_t NM = (size_t)N * M;
    for (int k = 0; k < CAO_F_COUNT; k++) {
        size_t w = k == CAO_F_OAS ? (size_t)e->K * 10 : (size_t)F_WIDTH[k];
        e->f[k] = (double*)calloc(NM * w + 1, sizeof(double));
    }
    for (int k = 0; k < CAO_U_COUNT; k++) {
        size_t n = k == CAO_U_GAME_OVER ? (size_t)N : (k == CAO_U_MAP ? (e->Kobs ? (size_t)N * MAPD * MAPD : 1) : NM);
        e->u[k] = (uint8_t*)calloc(n, 1);
    }
    for (int k = 0; k < CAO_I_COUNT; k++) e->i32[k] = (int32_t*)calloc(NM, sizeof(int32_t));
    e->goal = (double*)calloc(NM * 2, sizeof(double));
    e->ref_orth = (double*)calloc(NM * 2, sizeof(double));
    e->radius = (double*)calloc(NM, sizeof(double));
    e->pref_speed = (double*)calloc(NM, sizeof(double));
    e->ang_speed = (double*)calloc(NM, sizeof(double));
    e->cur_speed = (double*)calloc(NM, sizeof(double));
    e->cur_turn = (double*)calloc(NM, sizeof(double));
    e->coop = (double*)calloc(NM, sizeof(double));
    e->policy = (int32_t*)calloc(NM, sizeof(int32_t));
    e->dyn = (int32_t*)calloc(NM, sizeof(int32_t));
    e->nagents = (int32_t*)calloc(N, sizeof(int32_t));
    e->nobst = (int32_t*)calloc(N, sizeof(int32_t));
    e->sc_agents6 = (double*)calloc(NM * 6, sizeof(double));
    e->sc_heading0 = (double*)calloc(NM, sizeof(double));
    e->sc_coop = (double*)calloc(NM, sizeof(double));
    e->sc_obst = (double*)calloc((size_t)N * (e->Kobs ? e->Kobs : 1) * 4, sizeof(double));
    e->sc_policy = (int32_t*)calloc(NM, sizeof(int32_t));
    e->sc_dyn = (int32_t*)calloc(NM, sizeof(int32_t));
    e->sc_nagents = (int32_t*)calloc(N, sizeof(int32_t));
    e->sc_nobst = (int32_t*)calloc(N, sizeof(int32_t));
    e->sc_has_heading = (uint8_t*)calloc(1, 1);
    return e;
}

void cao_destroy(cao_env* e) {
    if (!e) return;
    for (int k = 0; k < CAO_F_COUNT; k++) free(e->f[k]);
    for (int k = 0; k < CAO_U_COUNT; k++) free(e->u[k]);
    for (int k = 0; k < CAO_I_COUNT; k++) free(e->i32[k]);
    free(e->goal); free(e->ref_orth); free(e->radius); free(e->pref_speed); free(e->ang_speed);
    free(e->cur_speed); free(e->cur_turn); free(e->coop); free(e->policy); free(e->dyn);
    free(e->nagents); free(e->nobst); free(e->sc_agents6); free(e->sc_heading0); free(e->sc_coop);
    free(e->sc_obst); free(e->sc_policy); free(e->sc_dyn); free(e->sc_nagents); free(e->sc_nobst);
    free(e->sc_has_heading);
    free(e);
}

double* cao_f64(cao_env* e, int field) { return e->f[field]; }
uint8_t* cao_u8(cao_env* e, int field) { return e->u[field]; }
int32_t* cao_i32(cao_env* e, int field) { return e->i32[field]; }

void cao_set_scenario(cao_env* e, const double* agents6, const double* heading0, const int32_t* policy_id,
                      const int32_t* dynamics_id, const int32_t* n_agents, const double* coop,
                      const double* obstacles, const int32_t* n_obst) {
    size_t NM = (size_t)e->N * e->M;
    memcpy(e->sc_agents6, agents6, NM * 6 * sizeof(double));
    e->sc_has_heading[0] = heading0 != NULL;
    if (heading0) memcpy(e->sc_heading0, heading0, NM * sizeof(double));
    memcpy(e->sc_policy, policy_id, NM * sizeof(int32_t));
    memcpy(e->sc_dyn, dynamics_id, NM * sizeof(int32_t));
    for (int w = 0; w < e->N; w++) e->sc_nagents[w] = n_agents ? n_agents[w] : e->M;
    for (size_t k = 0; k < NM; k++) e->sc_coop[k] = coop ? coop[k] : 1.0; /* agent.py:10 */
    for (int w = 0; w < e->N; w++) e->sc_nobst[w] = (n_obst && e->Kobs) ? n_obst[w] : 0;
    if (obstacles && e->Kobs) memcpy(e->sc_obst, obstacles, (size_t)e->N * e->Kobs * 4 * sizeof(double));
}

/* Map.world_coordinates_to_map_indices (Map.py:40-47) */
static int world_to_cell(double x, double y, int* gx, int* gy) {
    double cell = 0.1, ox = (30 / 2.) / cell, oy = (30 / 2.) / cell; /* Map.py:19 */
    *gx = (int)floor(ox - y / cell);
    *gy = (int)floor(oy + x / cell);
    return *gx >= 0 && *gy >= 0 && *gx < MAPD && *gy < MAPD;
}

/* Map.get_occupancy_grid (Map.py:107-123); obstacle corners [(xu,yu),(xl,yu),(xl,yl),(xu,yl)]
 * (test_cases.py:2496): filled inclusive from cell(corner[1]) to cell(corner[3]).  Python's
 * negative-index wrap is reproduced; indices >= 300 (IndexError in the reference) are skipped. */
void cao_rasterize(const double* obst, int n_obst, uint8_t* map) {
    memset(map, 0, MAPD * MAPD);
    for (int o = 0; o < n_obst; o++) {
        double xl = obst[o * 4 + 0], yl = obst[o * 4 + 1], xu = obst[o * 4 + 2], yu = obst[o * 4 + 3];
        int s0, s1, e0, e1;
        world_to_cell(xl, yu, &s0, &s1);
        world_to_cell(xu, yl, &e0, &e1);
        for (int ii = s0; ii <= e0; ii++)
            for (int jj = s1; jj <= e1; jj++) {
                int a = ii < 0 ? ii + MAPD : ii, b = jj < 0 ? jj + MAPD : jj;
                if (a < 0 || b < 0 || a >= MAPD || b >= MAPD) continue;
                map[a * MAPD + b] = 1;
            }
    }
}

/* Dynamics.update_ego_frame (dynamics/Dynamics.py:14-28) + Agent.get_ref (agent.py:250-269) */
static void update_ego_frame(cao_env* e, size_t a) {
    double* pos = e->f[CAO_F_POS] + 2 * a;
    double* goal = e->goal + 2 * a;
    double gx = goal[0] - pos[0], gy = goal[1] - pos[1];
    e->f[CAO_F_PAST_DIST_TO_GOAL][a] = e->f[CAO_F_DIST_TO_GOAL][a];
    double dist = sqrt(gx * gx + gy * gy);
    e->f[CAO_F_DIST_TO_GOAL][a] = dist;
    if (e->f[CAO_F_T][a] == 0) e->f[CAO_F_PAST_DIST_TO_GOAL][a] = dist;
    double px = gx, py = gy;
    if (dist > 1e-8) { px = gx / dist; py = gy / dist; }
    e->f[CAO_F_REF_PRLL][2 * a] = px;
    e->f[CAO_F_REF_PRLL][2 * a + 1] = py;
    e->ref_orth[2 * a] = -py;
    e->ref_orth[2 * a + 1] = px;
    double ang = atan2(py, px);
    double he = wrap(e->f[CAO_F_HEADING][a] - ang);
    e->f[CAO_F_HEADING_EGO][a] = he;
    double* vel = e->f[CAO_F_VEL] + 2 * a;
    double sp = sqrt(vel[0] * vel[0] + vel[1] * vel[1]);
    e->f[CAO_F_VEL_EGO][2 * a] = sp * cos(he);
    e->f[CAO_F_VEL_EGO][2 * a + 1] = sp * sin(he);
    e->f[CAO_F_REL_GOAL][2 * a] = goal[0] - pos[0];
    e->f[CAO_F_REL_GOAL][2 * a + 1] = goal[1] - pos[1];
}

/* Agent.__init__ (agent.py:9-109) */
static void init_agent(cao_env* e, int w, int i) {
    size_t a = (size_t)w * e->M + i;
    const double* s = e->sc_agents6 + a * 6;
    double* pos = e->f[CAO_F_POS] + 2 * a;
    pos[0] = s[0]; pos[1] = s[1];
    e->goal[2 * a] = s[2]; e->goal[2 * a + 1] = s[3];
    e->pref_speed[a] = s[4];
    e->radius[a] = s[5];
    e->f[CAO_F_VEL][2 * a] = e->f[CAO_F_VEL][2 * a + 1] = 0.0;
    e->f[CAO_F_SPEED][a] = 0.0;
    e->ang_speed[a] = 0.0;
    if (e->sc_has_heading[0]) e->f[CAO_F_HEADING][a] = e->sc_heading0[a];
    else e->f[CAO_F_HEADING][a] = atan2(s[3] - s[1], s[2] - s[0]); /* agent.py:29-31 */
    e->f[CAO_F_DELTA_HEADING][a] = 0.0;
    e->f[CAO_F_HEADING_EGO][a] = 0.0;
    e->f[CAO_F_VEL_EGO][2 * a] = e->f[CAO_F_VEL_EGO][2 * a + 1] = 0.0;
    memset(e->f[CAO_F_PAST_ACTIONS] + 4 * a, 0, 4 * sizeof(double));
    e->f[CAO_F_ACTION][2 * a] = e->f[CAO_F_ACTION][2 * a + 1] = 0.0;
    e->f[CAO_F_DIST_TO_GOAL][a] = 0.0;
    /* agent.py:59-63, config.py:60-61 */
    double straight = (norm2(pos[0] - s[2], pos[1] - s[3]) - 0.75) / s[4];
    e->f[CAO_F_TIME_REMAINING][a] = 3.0 * straight;
    e->f[CAO_F_T][a] = 0.0;
    e->i32[CAO_I_STEP_NUM][a] = 0;
    e->u[CAO_U_IS_AT_GOAL][a] = e->u[CAO_U_WAS_AT_GOAL][a] = 0;
    e->u[CAO_U_IN_COLLISION][a] = e->u[CAO_U_WAS_IN_COLLISION][a] = 0;
    e->u[CAO_U_RAN_OUT_OF_TIME][a] = e->u[CAO_U_IS_DONE][a] = 0;
    e->cur_speed[a] = e->cur_turn[a] = 0.0; /* UnicycleDynamicsMaxAcc.py:13-16 */
    e->policy[a] = e->sc_policy[a];
    e->dyn[a] = e->sc_dyn[a];
    e->coop[a] = e->sc_coop[a];
    e->f[CAO_F_REWARD][a] = 0.0;
    update_ego_frame(e, a); /* agent.py:92 */
}

/* OtherAgentsStatesSensor.sense (sensors/OtherAgentsStatesSensor.py:11-77) */
static void sense_oas(cao_env* e, int w, int i) {
    int M = e->M, K = e->K, n = e->nagents[w];
    size_t base = (size_t)w * M, a = base + i;
    double* out = e->f[CAO_F_OAS] + a * K * 10;
    memset(out, 0, (size_t)K * 10 * sizeof(double));
    int idx[64];
    double key[64];
    int cnt = 0;
    const double* P = e->f[CAO_F_POS];
    for (int j = 0; j < n; j++) {
        if (j == i) continue;
        double dx = P[2 * (base + j)] - P[2 * a], dy = P[2 * (base + j) + 1] - P[2 * a + 1];
        double dc = norm2(dx, dy);
        double d2 = dc - e->radius[a] - e->radius[base + j];
        /* SENSING_HORIZON = inf (config.py:63): never skipped */
        idx[cnt] = j; key[cnt] = d2; cnt++;
    }
    /* stable ascending insertion sort (Python sorted), then reverse, keep last K (:28-34) */
    for (int p = 1; p < cnt; p++) {
        int ji = idx[p]; double kk = key[p]; int q = p - 1;
        while (q >= 0 && key[q] > kk) { idx[q + 1] = idx[q]; key[q + 1] = key[q]; q--; }
        idx[q + 1] = ji; key[q + 1] = kk;
    }
    int order[64];
    for (int p = 0; p < cnt; p++) order[p] = idx[cnt - 1 - p];
    int start = cnt > K ? cnt - K : 0;
    int row = 0;
    const double* prll = e->f[CAO_F_REF_PRLL] + 2 * a;
    const double* orth = e->ref_orth + 2 * a;
    for (int p = start; p < cnt; p++, row++) {
        size_t b = base + order[p];
        double dx = P[2 * b] - P[2 * a], dy = P[2 * b + 1] - P[2 * a + 1];
        const double* v = e->f[CAO_F_VEL] + 2 * b;
        double* r = out + row * 10;
        r[0] = dx; r[1] = dy;
        r[2] = dot2(dx, dy, prll[0], prll[1]);
        r[3] = dot2(dx, dy, orth[0], orth[1]);
        r[4] = dot2(v[0], v[1], prll[0], prll[1]);
        r[5] = dot2(v[0], v[1], orth[0], orth[1]);
        r[6] = e->radius[b];
        r[7] = e->radius[a] + e->radius[b];
        r[8] = norm2(dx, dy) - e->radius[a] - e->radius[b];
        r[9] = e->policy[b] == CAO_POL_STATIC ? 1.0 : 2.0;
    }
    e->i32[CAO_I_NUM_OBSERVED][a] = row;
}

/* LaserScanSensor.sense (sensors/LaserScanSensor.py:9-22,27-58) with Map.py:49-79 */
static void sense_laser(cao_env* e, int w, int i) {
    size_t a = (size_t)w * e->M + i;
    double* out = e->f[CAO_F_LASERSCAN] + a * NBEAM;
    const uint8_t* map = e->nobst[w] > 0 ? e->u[CAO_U_MAP] + (size_t)w * MAPD * MAPD : NULL;
    double px = e->f[CAO_F_POS][2 * a], py = e->f[CAO_F_POS][2 * a + 1], h = e->f[CAO_F_HEADING][a];
    int egx, egy;
    int ego_in = world_to_cell(px, py, &egx, &egy); /* Map.get_agent_mask, Map.py:73-79 */
    double rr = e->radius[a] / 0.1;
    double r2 = rr * rr;
    double astep = (PI - (-PI)) / 15.0;     /* np.linspace(-pi, pi, 16) */
    double rstep = 2 * PI / 16;             /* range_resolution, LaserScanSensor.py:13 */
    for (int b = 0; b < NBEAM; b++) {
        double ang0 = b == NBEAM - 1 ? PI : (double)b * astep + (-PI);
        double ang = ang0 + h;
        double ca = cos(ang), sa = sin(ang);
        int count = 0, last = -1;
        for (int k = 0; k < NSAMP; k++) {
            double rg = 0.0 + (double)k * rstep;
            double x = px + rg * ca, y = py + rg * sa;
            int gx, gy;
            int in_map = world_to_cell(x, y, &gx, &gy);
            int hit = 0;
            if (in_map && map && map[gx * MAPD + gy]) {
                int masked = 0;
                if (ego_in) {
                    double dx = (double)(gy - egy), dy = (double)(gx - egx);
                    masked = dx * dx + dy * dy < r2; /* Map.get_agent_map_indices, Map.py:67-71 */
                }
                hit = !masked;
            }
            count += hit;
            if (count == 1) last = k; /* ranges[first_hits[0]] = ...: last k with cumsum == 1 wins (:43-47) */
        }
        double range = last >= 0 ? 0.0 + (double)last * rstep : 6.0;
        out[b] = 1 - range / 6;
    }
}

static void sense_world(cao_env* e, int w) {
    for (int i = 0; i < e->nagents[w]; i++) {
        sense_oas(e, w, i);
        if (e->cfg.laserscan) sense_laser(e, w, i);
    }
}

void cao_reset(cao_env* e, const uint8_t* world_mask) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int w = 0; w < e->N; w++) {
        if (world_mask && !world_mask[w]) continue;
        e->nagents[w] = e->sc_nagents[w];
        e->nobst[w] = e->sc_nobst[w];
        if (e->Kobs && e->nobst[w] > 0)
            cao_rasterize(e->sc_obst + (size_t)w * e->Kobs * 4, e->nobst[w], e->u[CAO_U_MAP] + (size_t)w * MAPD * MAPD);
        for (int i = 0; i < e->M; i++) init_agent(e, w, i);
        e->u[CAO_U_GAME_OVER][w] = 0;
        sense_world(e, w); /* env.py:266 */
    }
}

/* CARRLPolicy table (policies/CARRLPolicy.py:5-15): speed 1, dh = linspace(-pi/6, pi/6, 11) */
static double carrl_heading(int k) {
    double lo = -(PI / 6), hi = PI / 6;
    double step = (hi - lo) / 10.0;
    if (k == 10) return hi;
    return (double)k * step + lo;
}

/* ---------------- ORCA (RVO2 v2.0 Agent::computeNewVelocity) ---------------- */
typedef struct { float px, py, dx, dy; } orca_line;
#define RVO_EPS 0.00001f
static float detf(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

static int lp1(const orca_line* L, int no, float radius, float ox, float oy, int dir_opt, float* rx, float* ry) {
    float dot = L[no].px * L[no].dx + L[no].py * L[no].dy;
    float disc = dot * dot + radius * radius - (L[no].px * L[no].px + L[no].py * L[no].py);
    if (disc < 0.0f) return 0;
    float sq = sqrtf(disc);
    float tl = -dot - sq, tr = -dot + sq;
    for (int i = 0; i < no; i++) {
        float den = detf(L[no].dx, L[no].dy, L[i].dx, L[i].dy);
        float num = detf(L[i].dx, L[i].dy, L[no].px - L[i].px, L[no].py - L[i].py);
        if (fabsf(den) <= RVO_EPS) {
            if (num < 0.0f) return 0;
            continue;
        }
        float t = num / den;
        if (den >= 0.0f) tr = tr < t ? tr : t;
        else tl = tl > t ? tl : t;
        if (tl > tr) return 0;
    }
    float t;
    if (dir_opt) {
        t = (ox * L[no].dx + oy * L[no].dy > 0.0f) ? tr : tl;
    } else {
        t = L[no].dx * (ox - L[no].px) + L[no].dy * (oy - L[no].py);
        if (t < tl) t = tl;
        else if (t > tr) t = tr;
    }
    *rx = L[no].px + t * L[no].dx;
    *ry = L[no].py + t * L[no].dy;
    return 1;
}

static int lp2(const orca_line* L, int n, float radius, float ox, float oy, int dir_opt, float* rx, float* ry) {
    if (dir_opt) { *rx = ox * radius; *ry = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) {
        float inv = 1.0f / sqrtf(ox * ox + oy * oy); /* normalize(): Vector2 / abs */
        *rx = ox * inv * radius; *ry = oy * inv * radius;
    } else { *rx = ox; *ry = oy; }
    for (int i = 0; i < n; i++) {
        if (detf(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry) > 0.0f) {
            float tx = *rx, ty = *ry;
            if (!lp1(L, i, radius, ox, oy, dir_opt, rx, ry)) { *rx = tx; *ry = ty; return i; }
        }
    }
    return n;
}

/* linearProgram3(lines, numObstLines, beginLine, radius, result): the obstacle lines [0, num_obst) are hard
 * constraints, copied unprojected in front of the projected agent lines */
#define CAO_MAXLINES 192 /* 2 visible edges per rectangle x 64 rectangles + 63 neighbours, rounded */
static void lp3(const orca_line* L, int n, int num_obst, int begin, float radius, float* rx, float* ry) {
    float distance = 0.0f;
    orca_line P[CAO_MAXLINES];
    for (int i = begin; i < n; i++) {
        if (detf(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry) > distance) {
            int np = 0;
            for (int j = 0; j < num_obst; j++) P[np++] = L[j];
            for (int j = num_obst; j < i; j++) {
                orca_line ln;
                float d = detf(L[i].dx, L[i].dy, L[j].dx, L[j].dy);
                if (fabsf(d) <= RVO_EPS) {
                    if (L[i].dx * L[j].dx + L[i].dy * L[j].dy > 0.0f) continue;
                    ln.px = 0.5f * (L[i].px + L[j].px);
                    ln.py = 0.5f * (L[i].py + L[j].py);
                } else {
                    float s = detf(L[j].dx, L[j].dy, L[i].px - L[j].px, L[i].py - L[j].py) / d;
                    ln.px = L[i].px + s * L[i].dx;
                    ln.py = L[i].py + s * L[i].dy;
                }
                float ddx = L[j].dx - L[i].dx, ddy = L[j].dy - L[i].dy;
                float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
                ln.dx = ddx * inv; ln.dy = ddy * inv;
                P[np++] = ln;
            }
            float tx = *rx, ty = *ry;
            if (lp2(P, np, radius, -L[i].dy, L[i].dx, 1, rx, ry) < np) { *rx = tx; *ry = ty; }
            distance = detf(L[i].dx, L[i].dy, L[i].px - *rx, L[i].py - *ry);
        }
    }
}

/* ---- static obstacles in the ORCA solve ------------------------------------------------------------------------
 * RVOPolicy.find_next_action hands the world's rectangles to its private simulator (policies/RVOPolicy.py:56-57
 * sim.addObstacle, :45 sim.processObstacles at the first call only - later re-additions never reach the obstacle tree,
 * SURVEY Q21), timeHorizonObst = RVO_TIME_HORIZON (:25-28).  The arithmetic is RVO2 v2.0's (library absent: PARITY
 * UNPINNED, restated from the published algorithm): RVOSimulator::addObstacle (unit directions, convexity),
 * Agent::computeNeighbors / KdTree::queryObstacleTreeRecursive / Agent::insertObstacleNeighbor (an edge is a
 * neighbour when the agent is strictly on its right side and closer than timeHorizonObst * maxSpeed + radius;
 * nearest first) and the obstacle half of Agent::computeNewVelocity.
 * Stated deviation: RVO2 stores the edges in a BSP tree whose construction may SPLIT an edge that straddles the
 * supporting line of another edge (two collinear halves describing the same wall) and whose traversal fixes the order
 * of equidistant neighbours; here edges are never split and equidistant edges keep (rectangle, edge) index order -
 * the order RVO2 itself produces for a single rectangle.
 * A rectangle (xl, yl, xu, yu) is the counter-clockwise polygon [(xu,yu), (xl,yu), (xl,yl), (xu,yl)] of
 * test_cases.py:2496; Cython narrows the vertices to float. */
typedef struct { float x, y, ux, uy; int convex; } orca_vertex;

static void orca_rect_vertices(const double* rect, orca_vertex* V) {
    const float xl = (float)rect[0], yl = (float)rect[1], xu = (float)rect[2], yu = (float)rect[3];
    const float X[4] = {xu, xl, xl, xu}, Y[4] = {yu, yu, yl, yl};
    for (int k = 0; k < 4; k++) {
        const int nx = (k + 1) & 3, pv = (k + 3) & 3;
        const float ex = X[nx] - X[k], ey = Y[nx] - Y[k];
        const float inv = 1.0f / sqrtf(ex * ex + ey * ey); /* normalize(): Vector2 / abs */
        V[k].x = X[k]; V[k].y = Y[k];
        V[k].ux = ex * inv; V[k].uy = ey * inv;
        /* isConvex_ = leftOf(prev, this, next) >= 0;  leftOf(a, b, c) = det(a - c, b - a) */
        V[k].convex = detf(X[pv] - X[nx], Y[pv] - Y[nx], X[k] - X[pv], Y[k] - Y[pv]) >= 0.0f;
    }
}

/* distSqPointLineSegment(a, b, c) */
static float dist_sq_point_segment(float ax, float ay, float bx, float by, float cx, float cy) {
    const float r = ((cx - ax) * (bx - ax) + (cy - ay) * (by - ay)) / ((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
    if (r < 0.0f) return (cx - ax) * (cx - ax) + (cy - ay) * (cy - ay);
    if (r > 1.0f) return (cx - bx) * (cx - bx) + (cy - by) * (cy - by);
    const float qx = cx - (ax + r * (bx - ax)), qy = cy - (ay + r * (by - ay));
    return qx * qx + qy * qy;
}

/* One obstacle edge (vertex o1 -> its successor o2; pv = predecessor of o1) against the lines built so far.
 * Returns 1 and fills *out when the edge contributes a half-plane. */
static int orca_obstacle_line(const orca_vertex* o1, const orca_vertex* o2, const orca_vertex* pv, const orca_vertex* nx2,
                              float px, float py, float vx, float vy, float radius, float inv_tho,
                              const orca_line* L, int nl, orca_line* out) {
    (void)nx2;
    const float rp1x = o1->x - px, rp1y = o1->y - py, rp2x = o2->x - px, rp2y = o2->y - py;
    /* already covered by an earlier obstacle line? */
    for (int j = 0; j < nl; j++) {
        if (detf(inv_tho * rp1x - L[j].px, inv_tho * rp1y - L[j].py, L[j].dx, L[j].dy) - inv_tho * radius >= -RVO_EPS &&
            detf(inv_tho * rp2x - L[j].px, inv_tho * rp2y - L[j].py, L[j].dx, L[j].dy) - inv_tho * radius >= -RVO_EPS)
            return 0;
    }
    const float dsq1 = rp1x * rp1x + rp1y * rp1y, dsq2 = rp2x * rp2x + rp2y * rp2y;
    const float rsq = radius * radius;
    const float ovx = o2->x - o1->x, ovy = o2->y - o1->y;
    const float s = ((-rp1x) * ovx + (-rp1y) * ovy) / (ovx * ovx + ovy * ovy);
    const float lx = -rp1x - s * ovx, ly = -rp1y - s * ovy;
    const float dsq_line = lx * lx + ly * ly;
    orca_line ln;
    if (s < 0.0f && dsq1 <= rsq) { /* collision with the left vertex; ignored when non-convex */
        if (!o1->convex) return 0;
        const float nxv = -rp1y, nyv = rp1x, inv = 1.0f / sqrtf(nxv * nxv + nyv * nyv);
        ln.px = 0.0f; ln.py = 0.0f; ln.dx = nxv * inv; ln.dy = nyv * inv;
        *out = ln;
        return 1;
    } else if (s > 1.0f && dsq2 <= rsq) { /* collision with the right vertex; the neighbouring edge takes it otherwise */
        if (!(o2->convex && detf(rp2x, rp2y, o2->ux, o2->uy) >= 0.0f)) return 0;
        const float nxv = -rp2y, nyv = rp2x, inv = 1.0f / sqrtf(nxv * nxv + nyv * nyv);
        ln.px = 0.0f; ln.py = 0.0f; ln.dx = nxv * inv; ln.dy = nyv * inv;
        *out = ln;
        return 1;
    } else if (s >= 0.0f && s < 1.0f && dsq_line <= rsq) { /* collision with the segment */
        ln.px = 0.0f; ln.py = 0.0f; ln.dx = -o1->ux; ln.dy = -o1->uy;
        *out = ln;
        return 1;
    }
    /* no collision: legs */
    float llx, lly, rlx, rly; /* left / right leg direction */
    const orca_vertex *a1 = o1, *a2 = o2; /* obstacle1 / obstacle2 after the oblique-view substitutions */
    const orca_vertex* left_nb = pv;       /* obstacle1->prevObstacle_ */
    if (s < 0.0f && dsq_line <= rsq) { /* viewed obliquely: the left vertex defines the velocity obstacle */
        if (!o1->convex) return 0;
        a2 = o1;
        const float leg1 = sqrtf(dsq1 - rsq);
        llx = (rp1x * leg1 - rp1y * radius) / dsq1; lly = (rp1x * radius + rp1y * leg1) / dsq1;
        rlx = (rp1x * leg1 + rp1y * radius) / dsq1; rly = (-rp1x * radius + rp1y * leg1) / dsq1;
    } else if (s > 1.0f && dsq_line <= rsq) { /* viewed obliquely: the right vertex defines it */
        if (!o2->convex) return 0;
        a1 = o2;
        left_nb = o1; /* obstacle2->prevObstacle_ */
        const float leg2 = sqrtf(dsq2 - rsq);
        llx = (rp2x * leg2 - rp2y * radius) / dsq2; lly = (rp2x * radius + rp2y * leg2) / dsq2;
        rlx = (rp2x * leg2 + rp2y * radius) / dsq2; rly = (-rp2x * radius + rp2y * leg2) / dsq2;
    } else { /* usual situation */
        if (o1->convex) {
            const float leg1 = sqrtf(dsq1 - rsq);
            llx = (rp1x * leg1 - rp1y * radius) / dsq1; lly = (rp1x * radius + rp1y * leg1) / dsq1;
        } else { llx = -o1->ux; lly = -o1->uy; }
        if (o2->convex) {
            const float leg2 = sqrtf(dsq2 - rsq);
            rlx = (rp2x * leg2 + rp2y * radius) / dsq2; rly = (-rp2x * radius + rp2y * leg2) / dsq2;
        } else { rlx = o1->ux; rly = o1->uy; }
    }
    /* legs never point into a neighbouring edge of a convex vertex: the neighbour's cut-off line takes over */
    int left_foreign = 0, right_foreign = 0;
    if (a1->convex && detf(llx, lly, -left_nb->ux, -left_nb->uy) >= 0.0f) {
        llx = -left_nb->ux; lly = -left_nb->uy;
        left_foreign = 1;
    }
    if (a2->convex && detf(rlx, rly, a2->ux, a2->uy) <= 0.0f) {
        rlx = a2->ux; rly = a2->uy;
        right_foreign = 1;
    }
    /* cut-off centres */
    const float lcx = inv_tho * (a1->x - px), lcy = inv_tho * (a1->y - py);
    const float rcx = inv_tho * (a2->x - px), rcy = inv_tho * (a2->y - py);
    const float cvx = rcx - lcx, cvy = rcy - lcy;
    const int same = a1 == a2;
    const float t = same ? 0.5f : ((vx - lcx) * cvx + (vy - lcy) * cvy) / (cvx * cvx + cvy * cvy);
    const float t_left = (vx - lcx) * llx + (vy - lcy) * lly;
    const float t_right = (vx - rcx) * rlx + (vy - rcy) * rly;
    if ((t < 0.0f && t_left < 0.0f) || (same && t_left < 0.0f && t_right < 0.0f)) { /* left cut-off circle */
        const float wx = vx - lcx, wy = vy - lcy, inv = 1.0f / sqrtf(wx * wx + wy * wy);
        const float uwx = wx * inv, uwy = wy * inv;
        ln.dx = uwy; ln.dy = -uwx;
        ln.px = lcx + radius * inv_tho * uwx; ln.py = lcy + radius * inv_tho * uwy;
        *out = ln;
        return 1;
    } else if (t > 1.0f && t_right < 0.0f) { /* right cut-off circle */
        const float wx = vx - rcx, wy = vy - rcy, inv = 1.0f / sqrtf(wx * wx + wy * wy);
        const float uwx = wx * inv, uwy = wy * inv;
        ln.dx = uwy; ln.dy = -uwx;
        ln.px = rcx + radius * inv_tho * uwx; ln.py = rcy + radius * inv_tho * uwy;
        *out = ln;
        return 1;
    }
    /* left leg, right leg or cut-off line, whichever is closest to the velocity */
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
    if (dc <= dl && dc <= dr) { /* cut-off line */
        ln.dx = -a1->ux; ln.dy = -a1->uy;
        ln.px = lcx + radius * inv_tho * (-ln.dy); ln.py = lcy + radius * inv_tho * ln.dx;
        *out = ln;
        return 1;
    } else if (dl <= dr) { /* left leg */
        if (left_foreign) return 0;
        ln.dx = llx; ln.dy = lly;
        ln.px = lcx + radius * inv_tho * (-ln.dy); ln.py = lcy + radius * inv_tho * ln.dx;
        *out = ln;
        return 1;
    }
    if (right_foreign) return 0; /* right leg */
    ln.dx = -rlx; ln.dy = -rly;
    ln.px = rcx + radius * inv_tho * (-ln.dy); ln.py = rcy + radius * inv_tho * ln.dx;
    *out = ln;
    return 1;
}

/* Obstacle neighbours (nearest first) and their ORCA lines; returns numObstLines (<= 2 per rectangle for an agent
 * outside it, 0 for one inside). */
static int orca_obstacle_lines(const double* rects, int n_obst, float px, float py, float vx, float vy, float radius,
                               float max_speed, float time_horizon_obst, orca_line* L) {
    int nb_rect[4 * CAO_MAXOBST], nb_edge[4 * CAO_MAXOBST], nn = 0;
    float nb_d[4 * CAO_MAXOBST];
    const float range = time_horizon_obst * max_speed + radius, range_sq = range * range;
    orca_vertex V[4];
    if (n_obst > CAO_MAXOBST) n_obst = CAO_MAXOBST;
    for (int r = 0; r < n_obst; r++) {
        orca_rect_vertices(rects + 4 * r, V);
        for (int k = 0; k < 4; k++) {
            const orca_vertex *o1 = &V[k], *o2 = &V[(k + 1) & 3];
            /* agentLeftOfLine = leftOf(o1, o2, position) = det(o1 - position, o2 - o1) */
            const float left = detf(o1->x - px, o1->y - py, o2->x - o1->x, o2->y - o1->y);
            const float ex = o2->x - o1->x, ey = o2->y - o1->y;
            const float dsq_line = (left * left) / (ex * ex + ey * ey);
            if (!(dsq_line < range_sq) || !(left < 0.0f)) continue; /* only from its right side (the agent can see it) */
            const float dsq = dist_sq_point_segment(o1->x, o1->y, o2->x, o2->y, px, py);
            if (!(dsq < range_sq)) continue;
            int i = nn++; /* Agent::insertObstacleNeighbor: insertion sort, strict < */
            while (i != 0 && dsq < nb_d[i - 1]) { nb_d[i] = nb_d[i - 1]; nb_rect[i] = nb_rect[i - 1]; nb_edge[i] = nb_edge[i - 1]; i--; }
            nb_d[i] = dsq; nb_rect[i] = r; nb_edge[i] = k;
        }
    }
    const float inv_tho = 1.0f / time_horizon_obst;
    int nl = 0;
    for (int q = 0; q < nn; q++) {
        orca_rect_vertices(rects + 4 * nb_rect[q], V);
        const int k = nb_edge[q];
        orca_line ln;
        if (orca_obstacle_line(&V[k], &V[(k + 1) & 3], &V[(k + 3) & 3], &V[(k + 2) & 3], px, py, vx, vy, radius, inv_tho, L, nl, &ln))
            L[nl++] = ln;
    }
    return nl;
}

/* ---- the LIBRARY half: what rvo2.PyRVOSimulator.doStep() computes for one agent of the simulator ----------------
 * (RVO2 v2.0 Agent::computeNeighbors + computeNewVelocity + update; library absent: PARITY UNPINNED).
 * All inputs are what the Python side handed over through the simulator's setters, already narrowed to float by the
 * Cython binding.  Neighbours are visited in index order (the real library's k-d tree order is unpinned), nearest
 * first, at most max_neighbors, closer than neighbor_dist.  collab: line.point = v + collab*u (assumed semantics of
 * the mit-acl fork's setAgentCollabCoeff; stock RVO2 uses 0.5).  rects [n_obst][4] = xl, yl, xu, yu (may be NULL):
 * obstacle lines come first and are hard constraints in linearProgram3.
 * tests/golden/rvo2_standin.py calls THIS function from inside the unmodified reference (RVOPolicy.py:88), so the
 * reference-run fixtures pin everything of RVOPolicy.py:53-117 around it. */
void cao_rvo2_step_agent(int n, int ego, const float* pos, const float* vel, const float* radius,
                         const float* pref_vel2, float max_speed, float collab, float neighbor_dist, int max_neighbors,
                         float time_horizon, float time_horizon_obst, float time_step, const double* rects, int n_obst,
                         float* new_pos2, float* new_vel2, float* lines_out, int* n_lines_out) {
    const float (*p)[2] = (const float (*)[2])pos;
    const float (*v)[2] = (const float (*)[2])vel;
    const float* r = radius;
    const float pvx = pref_vel2[0], pvy = pref_vel2[1];
    /* neighbour list: insertion by distSq, strict <, at most maxNeighbors (Agent::insertAgentNeighbor) */
    int nb[64]; float nd[64]; int nn = 0;
    if (n > 64) n = 64;
    if (max_neighbors < 0) max_neighbors = 0;
    if (max_neighbors > 63) max_neighbors = 63;
    float range_sq = neighbor_dist * neighbor_dist; /* rangeSq = sqr(neighborDist_) */
    for (int a = 0; a < n; a++) {
        if (a == ego) continue;
        float dx = p[ego][0] - p[a][0], dy = p[ego][1] - p[a][1];
        float dsq = dx * dx + dy * dy;
        if (max_neighbors > 0 && dsq < range_sq) {
            if (nn < max_neighbors) nn++;
            int i = nn - 1;
            while (i != 0 && dsq < nd[i - 1]) { nb[i] = nb[i - 1]; nd[i] = nd[i - 1]; i--; }
            nb[i] = a; nd[i] = dsq;
            if (nn == max_neighbors) range_sq = nd[nn - 1];
        }
    }
    orca_line Lall[CAO_MAXLINES];
    const int num_obst = (rects && n_obst > 0)
        ? orca_obstacle_lines(rects, n_obst, p[ego][0], p[ego][1], v[ego][0], v[ego][1], r[ego], max_speed, time_horizon_obst, Lall)
        : 0;
    orca_line* L = Lall + num_obst;
    float inv_th = 1.0f / time_horizon;
    float c = collab;
    for (int k = 0; k < nn; k++) {
        int o = nb[k];
        float rpx = p[o][0] - p[ego][0], rpy = p[o][1] - p[ego][1];
        float rvx = v[ego][0] - v[o][0], rvy = v[ego][1] - v[o][1];
        float dsq = rpx * rpx + rpy * rpy;
        float cr = r[ego] + r[o], crsq = cr * cr;
        float ux, uy;
        orca_line ln;
        if (dsq > crsq) {
            float wx = rvx - inv_th * rpx, wy = rvy - inv_th * rpy;
            float wlsq = wx * wx + wy * wy;
            float dp1 = wx * rpx + wy * rpy;
            if (dp1 < 0.0f && dp1 * dp1 > crsq * wlsq) {
                float wl = sqrtf(wlsq);
                float inv = 1.0f / wl;
                float uwx = wx * inv, uwy = wy * inv;
                ln.dx = uwy; ln.dy = -uwx;
                float s = cr * inv_th - wl;
                ux = s * uwx; uy = s * uwy;
            } else {
                float leg = sqrtf(dsq - crsq);
                float inv = 1.0f / dsq;
                if (detf(rpx, rpy, wx, wy) > 0.0f) {
                    ln.dx = (rpx * leg - rpy * cr) * inv;
                    ln.dy = (rpx * cr + rpy * leg) * inv;
                } else {
                    ln.dx = -((rpx * leg + rpy * cr) * inv);
                    ln.dy = -((-rpx * cr + rpy * leg) * inv);
                }
                float dp2 = rvx * ln.dx + rvy * ln.dy;
                ux = dp2 * ln.dx - rvx; uy = dp2 * ln.dy - rvy;
            }
        } else {
            float inv_ts = 1.0f / time_step;
            float wx = rvx - inv_ts * rpx, wy = rvy - inv_ts * rpy;
            float wl = sqrtf(wx * wx + wy * wy);
            float inv = 1.0f / wl;
            float uwx = wx * inv, uwy = wy * inv;
            ln.dx = uwy; ln.dy = -uwx;
            float s = cr * inv_ts - wl;
            ux = s * uwx; uy = s * uwy;
        }
        ln.px = v[ego][0] + c * ux;
        ln.py = v[ego][1] + c * uy;
        L[k] = ln;
    }
    float nvx, nvy;
    const int nl = num_obst + nn;
    int fail = lp2(Lall, nl, max_speed, pvx, pvy, 0, &nvx, &nvy);
    if (fail < nl) lp3(Lall, nl, num_obst, fail, max_speed, &nvx, &nvy);
    if (new_vel2) { new_vel2[0] = nvx; new_vel2[1] = nvy; }
    if (n_lines_out) { n_lines_out[0] = num_obst; n_lines_out[1] = nl; }
    if (lines_out)
        for (int k = 0; k < nl; k++) {
            lines_out[4 * k] = Lall[k].px; lines_out[4 * k + 1] = Lall[k].py;
            lines_out[4 * k + 2] = Lall[k].dx; lines_out[4 * k + 3] = Lall[k].dy;
        }
    /* Agent::update: position_ += velocity_ * timeStep_ (fp32) */
    new_pos2[0] = p[ego][0] + nvx * time_step;
    new_pos2[1] = p[ego][1] + nvy * time_step;
}

/* ---- the PYTHON half before doStep(): what RVOPolicy.find_next_action hands to its private simulator ---------------
 * (policies/RVOPolicy.py:63-85; simulator parameters :25-28: timeStep = Config.DT, neighborDist = SENSING_HORIZON = inf,
 * maxNeighbors = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT (:15), timeHorizon = timeHorizonObst = RVO_TIME_HORIZON = 5).
 * Every agent of the env (done or not, any policy) is an agent of the simulator: position, velocity, 1.15 * radius
 * (:76), maxSpeed = pref_speed (:75); the ego's preferred velocity pref_speed / ||g - p|| * (g - p) in fp64 (:71-72);
 * the ego's cooperation_coef (:85).  Narrowed to float where the Cython binding narrows.  PINNED by the setter
 * arguments recorded in tests/golden/rvo_episodes.npz. */
void cao_rvo_sim_inputs(int M, int ego, const double* pos, const double* vel, const double* goal,
                        const double* pref_speed, const double* radius, double collab,
                        float* p32, float* v32, float* r32, float* pref_vel2, float* max_speed, float* collab32) {
    for (int a = 0; a < M; a++) {
        p32[2 * a] = (float)pos[2 * a]; p32[2 * a + 1] = (float)pos[2 * a + 1];
        v32[2 * a] = (float)vel[2 * a]; v32[2 * a + 1] = (float)vel[2 * a + 1];
        r32[a] = (float)((1 + 15e-2) * radius[a]);
    }
    double gx = goal[2 * ego] - pos[2 * ego], gy = goal[2 * ego + 1] - pos[2 * ego + 1];
    double sc = pref_speed[ego] / norm2(gx, gy);
    pref_vel2[0] = (float)(sc * gx); pref_vel2[1] = (float)(sc * gy);
    *max_speed = (float)pref_speed[ego];
    *collab32 = (float)collab;
}

/* RVOPolicy.find_next_action (policies/RVOPolicy.py:53-117), ego LP only = the two Python halves around
 * the library call.  new_vel_out (optional): the fp32 velocity the linear programs chose, lines_out / n_lines_out
 * (optional): the half-planes in solve order and {numObstLines, total}. */
void cao_orca_action_ex(int M, int ego, const double* pos, const double* vel, const double* goal,
                        const double* pref_speed, const double* radius, double heading, double collab,
                        double dt, int max_neighbors, const double* rects, int n_obst, double* action_out,
                        float* new_vel_out, float* lines_out, int* n_lines_out) {
    float p[64][2], v[64][2], r[64], pv[2], max_speed, c, np2[2];
    cao_rvo_sim_inputs(M, ego, pos, vel, goal, pref_speed, radius, collab, &p[0][0], &v[0][0], r, pv, &max_speed, &c);
    cao_rvo2_step_agent(M, ego, &p[0][0], &v[0][0], r, pv, max_speed, c, INFINITY, max_neighbors, 5.0f, 5.0f, (float)dt,
                        rects, n_obst, np2, new_vel_out, lines_out, n_lines_out);
    /* back in Python, fp64 (RVOPolicy.py:91-106) */
    double dpx = (double)np2[0] - pos[2 * ego], dpy = (double)np2[1] - pos[2 * ego + 1];
    double ang1 = atan2(dpy, dpx);
    /* (ang1 - 0) % (2*pi), Python float modulo; |ang1| <= pi so fmod(ang1, 2*pi) == ang1 exactly */
    double nh = ang1;
    if (nh < 0) nh += 2 * PI;
    double dh = wrap(nh - heading);
    double speed = 1 / dt * norm2(dpx, dpy);
    if (fabs(dh) > PI / 6) {
        dh = (dh > 0 ? 1.0 : (dh < 0 ? -1.0 : 0.0)) * (PI / 6);
        speed = 0.;
    }
    action_out[0] = speed;
    action_out[1] = dh;
}

/* free space, maxNeighbors 10: the entry point the property tests of round 1 use */
void cao_orca_action(int M, int ego, const double* pos, const double* vel, const double* goal,
                     const double* pref_speed, const double* radius, double heading, double collab,
                     double dt, double* action_out) {
    cao_orca_action_ex(M, ego, pos, vel, goal, pref_speed, radius, heading, collab, dt, 10, NULL, 0, action_out, NULL, NULL, NULL);
}

/* _take_action policy dispatch (env.py:287-340) for one agent: returns the fp32-rounded action */
static void select_action(cao_env* e, int w, int i, const double* ext, float* act) {
    size_t base = (size_t)w * e->M, a = base + i;
    double a0 = 0.0, a1 = 0.0;
    act[0] = act[1] = 0.0f; /* env.py:289 */
    if (e->u[CAO_U_IS_DONE][a]) return; /* env.py:299-300 */
    switch (e->policy[a]) {
        case CAO_POL_STATIC: a0 = 0.0; a1 = 0.0; break; /* StaticPolicy.py:9-12 */
        case CAO_POL_NONCOOP: /* NonCooperativePolicy.py:10-13 */
            a0 = e->pref_speed[a]; a1 = -e->f[CAO_F_HEADING_EGO][a]; break;
        case CAO_POL_EXTERNAL: case CAO_POL_IGMCTS: case CAO_POL_GA3C:
            a0 = ext ? ext[2 * a] : 0.0; a1 = ext ? ext[2 * a + 1] : 0.0; break;
        case CAO_POL_LEARNING: /* LearningPolicy.py:11-16, max_heading_change = 4 (env.py:97,466-468) */
            a1 = 4.0 * (2. * (ext ? ext[2 * a + 1] : 0.0) - 1.);
            a0 = e->pref_speed[a] * (ext ? ext[2 * a] : 0.0);
            break;
        case CAO_POL_CARRL: {
            int k = ext ? (int)ext[2 * a] : 0;
            a0 = 1.0; a1 = carrl_heading(k); break;
        }
        case CAO_POL_RVO: {
            double out[2];
            int n = e->nagents[w];
            const int maxnb = e->cfg.rvo_max_neighbors > 0 ? e->cfg.rvo_max_neighbors : e->M; /* RVOPolicy.py:15 */
            cao_orca_action_ex(n, i, e->f[CAO_F_POS] + 2 * base, e->f[CAO_F_VEL] + 2 * base, e->goal + 2 * base,
                               e->pref_speed + base, e->radius + base, e->f[CAO_F_HEADING][a], e->coop[a],
                               e->cfg.dt, maxnb, e->Kobs ? e->sc_obst + (size_t)w * e->Kobs * 4 : NULL,
                               e->Kobs ? e->nobst[w] : 0, out, NULL, NULL, NULL);
            a0 = out[0]; a1 = out[1]; break;
        }
    }
    act[0] = (float)a0;
    act[1] = (float)a1;
}

/* Agent.take_action (agent.py:147-190) with the dynamics models (dynamics/*.py) */
static void take_action(cao_env* e, size_t a, const float* actf) {
    double dt = e->cfg.dt;
    uint8_t* U = e->u[CAO_U_IS_AT_GOAL];
    if (U[a] || e->u[CAO_U_RAN_OUT_OF_TIME][a] || e->u[CAO_U_IN_COLLISION][a]) { /* agent.py:148-159 */
        if (U[a]) e->u[CAO_U_WAS_AT_GOAL][a] = 1;
        if (e->u[CAO_U_IN_COLLISION][a]) e->u[CAO_U_WAS_IN_COLLISION][a] = 1;
        if (!U[a]) e->f[CAO_F_T][a] += dt;
        e->f[CAO_F_VEL][2 * a] = e->f[CAO_F_VEL][2 * a + 1] = 0.0;
        return;
    }
    double a0 = (double)actf[0], a1 = (double)actf[1];
    double* pa = e->f[CAO_F_PAST_ACTIONS] + 4 * a; /* agent.py:162-163 */
    pa[2] = pa[0]; pa[3] = pa[1]; pa[0] = a0; pa[1] = a1;
    double h = e->f[CAO_F_HEADING][a];
    double speed, hn;
    double* vel = e->f[CAO_F_VEL] + 2 * a;
    switch (e->dyn[a]) {
        default:
        case CAO_DYN_UNICYCLE: /* UnicycleDynamics.py:10-24 */
            speed = a0; hn = wrap(a1 + h); break;
        case CAO_DYN_MAXTURNRATE: { /* UnicycleDynamicsMaxTurnRate.py:11-25 */
            double tr = clipd(a1 / dt, -3.0, 3.0);
            speed = a0; hn = wrap(tr * dt + h); break;
        }
        case CAO_DYN_MAXACC: { /* UnicycleDynamicsMaxAcc.py:17-39 */
            double tr = clipd(a1 / dt, -3.0, 3.0);
            double lacc = clipd(2.0 * (a0 - e->cur_speed[a]), -2.0, 2.0);
            double tacc = clipd(2.0 * (tr - e->cur_turn[a]), -3.0, 3.0);
            e->cur_speed[a] += lacc * dt;
            e->cur_speed[a] = clipd(e->cur_speed[a], -1.1, 1.1);
            e->cur_turn[a] += tacc * dt;
            speed = e->cur_speed[a]; hn = wrap(e->cur_turn[a] * dt + h); break;
        }
        case CAO_DYN_SECONDORDER: { /* UnicycleSecondOrderEulerDynamics.py:12-29 */
            speed = clipd(norm2(vel[0], vel[1]) + a0 * dt, 0.0, 1.0);
            double tr = e->ang_speed[a] + a1 * dt;
            e->ang_speed[a] = clipd(tr, -3.0, 3.0);
            hn = wrap(e->ang_speed[a] * dt + h); break;
        }
        case CAO_DYN_FIRSTORDER: /* FirstOrderDynamics.py:10-23 */
            speed = a0; hn = wrap(a1 * dt + h); break;
    }
    double c = cos(hn), s = sin(hn);
    double dx = speed * c * dt, dy = speed * s * dt;
    e->f[CAO_F_POS][2 * a] += dx;
    e->f[CAO_F_POS][2 * a + 1] += dy;
    vel[0] = speed * c; vel[1] = speed * s;
    e->f[CAO_F_SPEED][a] = speed;
    e->f[CAO_F_DELTA_HEADING][a] = wrap(hn - h);
    e->f[CAO_F_HEADING][a] = hn;
    update_ego_frame(e, a);
    /* end_conditions._check_if_at_goal (utils/end_conditions.py:3-6) */
    double ex = e->f[CAO_F_POS][2 * a] - e->goal[2 * a], ey = e->f[CAO_F_POS][2 * a + 1] - e->goal[2 * a + 1];
    U[a] = ex * ex + ey * ey <= 0.75 * 0.75;
    e->f[CAO_F_TIME_REMAINING][a] -= dt; /* agent.py:184-188 */
    e->f[CAO_F_T][a] += dt;
    e->i32[CAO_I_STEP_NUM][a] += 1;
    if (e->f[CAO_F_TIME_REMAINING][a] <= 0.0) e->u[CAO_U_RAN_OUT_OF_TIME][a] = 1;
}

/* _compute_rewards + _check_for_collisions (env.py:502-567, 630-671) */
static void rewards_world(cao_env* e, int w) {
    int M = e->M, n = e->nagents[w];
    size_t base = (size_t)w * M;
    uint8_t coll_agent[64] = {0}, coll_wall[64] = {0};
    double dmin[64];
    for (int i = 0; i < n; i++) dmin[i] = INFINITY;
    const double* P = e->f[CAO_F_POS];
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) {
            if (e->policy[base + j] == CAO_POL_STATIC && !e->cfg.collide_with_static) continue; /* env.py:643 (Q8) */
            double d = norm2(P[2 * (base + i)] - P[2 * (base + j)], P[2 * (base + i) + 1] - P[2 * (base + j) + 1]);
            double cr = e->radius[base + i] + e->radius[base + j];
            double g = d - cr;
            if (g < dmin[i]) dmin[i] = g; /* index i only (Q7) */
            if (d <= cr) coll_agent[i] = coll_agent[j] = 1;
        }
    if (e->nobst[w] > 0) { /* env.py:656-666 */
        const uint8_t* map = e->u[CAO_U_MAP] + (size_t)w * MAPD * MAPD;
        for (int i = 0; i < n; i++) {
            int pi, pj;
            int in_map = world_to_cell(P[2 * (base + i)], P[2 * (base + i) + 1], &pi, &pj);
            if (!in_map) continue;
            double rr = e->radius[base + i] / 0.1, r2 = rr * rr;
            int R = (int)ceil(rr) + 1, hit = 0;
            for (int y = pi - R; y <= pi + R && !hit; y++)
                for (int x = pj - R; x <= pj + R; x++) {
                    if (y < 0 || x < 0 || y >= MAPD || x >= MAPD) continue;
                    double dx = (double)(x - pj), dy = (double)(y - pi);
                    if (dx * dx + dy * dy < r2 && map[y * MAPD + x]) { hit = 1; break; }
                }
            coll_wall[i] = (uint8_t)hit;
        }
    }
    for (int i = 0; i < n; i++) {
        size_t a = base + i;
        double r = -0.01;
        if (e->u[CAO_U_IS_AT_GOAL][a]) {
            if (!e->u[CAO_U_WAS_AT_GOAL][a]) r = 3.0;
        } else {
            if (!e->u[CAO_U_WAS_IN_COLLISION][a]) {
                if (coll_agent[i]) { r = -10.0; e->u[CAO_U_IN_COLLISION][a] = 1; }
                else if (coll_wall[i]) { r = -0.25; e->u[CAO_U_IN_COLLISION][a] = 1; }
                else {
                    if (dmin[i] <= 0.2) r += -0.1 - dmin[i] / 2.;
                    /* wiggly-behaviour term: threshold 0.0, value 0.0 (config.py:41-42) -> += 0.0 */
                    const double* pa = e->f[CAO_F_PAST_ACTIONS] + 4 * a;
                    if (norm2(pa[2] - pa[0], pa[3] - pa[1]) > 0.0) r += 0.0;
                }
            } else if (e->u[CAO_U_RAN_OUT_OF_TIME][a]) {
                r += -10.0; /* Q9: reachable only after a collision */
            }
            r += 0.0 * (e->f[CAO_F_PAST_DIST_TO_GOAL][a] - e->f[CAO_F_DIST_TO_GOAL][a]); /* env.py:561 */
        }
        r = clipd(r, -10.0, 3.0) / (3.0 - (-10.0)); /* env.py:563-564 */
        e->f[CAO_F_REWARD][a] = r;
    }
}

/* _check_which_agents_done (env.py:711-738) */
static void done_world(cao_env* e, int w) {
    int n = e->nagents[w];
    size_t base = (size_t)w * e->M;
    int all = 1, all_learning = 1;
    for (int i = 0; i < n; i++) {
        size_t a = base + i;
        uint8_t d = e->u[CAO_U_IS_AT_GOAL][a] | e->u[CAO_U_RAN_OUT_OF_TIME][a] | e->u[CAO_U_IN_COLLISION][a];
        e->u[CAO_U_IS_DONE][a] = d;
        all &= d;
        if (e->policy[a] == CAO_POL_LEARNING) all_learning &= d;
    }
    uint8_t go;
    switch (e->cfg.game_over_mode) {
        case CAO_GO_ALL: go = (uint8_t)all; break;
        case CAO_GO_LEARNING: go = (uint8_t)all_learning; break;
        default: go = n > 0 ? e->u[CAO_U_IS_DONE][base] : 1; break;
    }
    e->u[CAO_U_GAME_OVER][w] = go;
}

void cao_step(cao_env* e, const double* ext_actions) {
    /* worlds are independent (no cross-world term in env.py): with -fopenmp the loop is shared between threads, used
     * only by bench.py's cpu_baseline (OMP_NUM_THREADS); results do not depend on the thread count */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int w = 0; w < e->N; w++) {
        float act[64][2];
        int n = e->nagents[w];
        size_t base = (size_t)w * e->M;
        for (int i = 0; i < n; i++) select_action(e, w, i, ext_actions, act[i]); /* all select first ... */
        for (int i = 0; i < n; i++) {                                            /* ... then all move (env.py:334-335) */
            e->f[CAO_F_ACTION][2 * (base + i)] = act[i][0];
            e->f[CAO_F_ACTION][2 * (base + i) + 1] = act[i][1];
            take_action(e, base + i, act[i]);
        }
        rewards_world(e, w);
        sense_world(e, w);
        done_world(e, w);
    }
}

/* bench.py's cpu_baseline: fix the size of the thread team the world loops use (the OMP_NUM_THREADS variable is read
 * only when libgomp starts, i.e. before bench.py could set it) and report the size actually in force. */
int cao_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* bench.py's cpu_baseline loop in one call: n_steps x { step every world; restart the worlds whose game is over on the
 * same scenario }, i.e. what the DummyVecEnv loop of experiments/src/env_utils.py:29-62 does per env. */
void cao_run(cao_env* e, int n_steps) {
    for (int t = 0; t < n_steps; t++) {
        cao_step(e, NULL);
        int any = 0;
        for (int w = 0; w < e->N; w++) any |= e->u[CAO_U_GAME_OVER][w];
        if (any) cao_reset(e, e->u[CAO_U_GAME_OVER]);
    }
}

/* GA3CCADRLPolicy.agents_to_ga3c_cadrl_state (policies/GA3CCADRLPolicy.py:45-106), RNN architecture:
 * out[N,M,76] = [id, n_others, dist_to_goal, heading_ego, pref_speed, radius, 10 x (p_prll, p_orth, v_prll,
 * v_orth, r_other, r_host + r_other, edge distance)]; others sorted by (-round(d, 2), p_orth) (stable), the
 * last max_observed kept (farthest first, closest last).  round() is NumPy's: rint(x * 100) / 100. */
void cao_ga3c_states(cao_env* e, int max_observed, double* out) {
    int M = e->M;
    memset(out, 0, (size_t)e->N * M * 76 * sizeof(double));
    for (int w = 0; w < e->N; w++) {
        int n = e->nagents[w];
        size_t base = (size_t)w * M;
        for (int i = 0; i < n; i++) {
            size_t a = base + i;
            double* o = out + a * 76;
            const double* P = e->f[CAO_F_POS];
            const double* prll = e->f[CAO_F_REF_PRLL] + 2 * a;
            const double* orth = e->ref_orth + 2 * a;
            int idx[64], cnt = 0;
            double k1[64], k2[64];
            for (int j = 0; j < n; j++) {
                if (j == i) continue;
                double dx = P[2 * (base + j)] - P[2 * a], dy = P[2 * (base + j) + 1] - P[2 * a + 1];
                double d2 = norm2(dx, dy) - e->radius[a] - e->radius[base + j];
                idx[cnt] = j;
                k1[cnt] = -(rint(d2 * 100.0) / 100.0);
                k2[cnt] = dot2(dx, dy, orth[0], orth[1]);
                cnt++;
            }
            for (int p = 1; p < cnt; p++) { /* stable insertion sort by (k1, k2) */
                int ji = idx[p]; double a1 = k1[p], a2 = k2[p]; int q = p - 1;
                while (q >= 0 && (k1[q] > a1 || (k1[q] == a1 && k2[q] > a2))) {
                    idx[q + 1] = idx[q]; k1[q + 1] = k1[q]; k2[q + 1] = k2[q]; q--;
                }
                idx[q + 1] = ji; k1[q + 1] = a1; k2[q + 1] = a2;
            }
            int start = cnt > max_observed ? cnt - max_observed : 0, row = 0;
            o[0] = (double)i;
            o[2] = e->f[CAO_F_DIST_TO_GOAL][a];
            o[3] = e->f[CAO_F_HEADING_EGO][a];
            o[4] = e->pref_speed[a];
            o[5] = e->radius[a];
            for (int p = start; p < cnt; p++, row++) {
                size_t b = base + idx[p];
                double dx = P[2 * b] - P[2 * a], dy = P[2 * b + 1] - P[2 * a + 1];
                const double* v = e->f[CAO_F_VEL] + 2 * b;
                double* r = o + 6 + 7 * row;
                r[0] = dot2(dx, dy, prll[0], prll[1]);
                r[1] = dot2(dx, dy, orth[0], orth[1]);
                r[2] = dot2(v[0], v[1], prll[0], prll[1]);
                r[3] = dot2(v[0], v[1], orth[0], orth[1]);
                r[4] = e->radius[b];
                r[5] = e->radius[a] + e->radius[b];
                r[6] = norm2(dx, dy) - e->radius[a] - e->radius[b];
            }
            o[1] = (double)row;
        }
    }
}
