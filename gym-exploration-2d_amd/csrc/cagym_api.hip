// cagym_api.hip -- C ABI of libcagym_hip.so (include/cagym.h): handle, device buffers, launches.
// No torch types, no CPU fallback: without a HIP device cagym_create fails.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cagym.h"
#include "cagym_kernels.h"
#include "cagym_kernels3.h"  // LDS layout helpers; the kernels themselves are instantiated in the cagym_k3_tu.hip units
#include "cagym_split3.h"
#include "cagym_launch3.h"
#ifdef CAGYM_MONOLITHIC  // diagnostic builds: every generation-3 specialisation in this one translation unit
#include "cagym_k3_all.inc"
#endif
#include "cagym_ig.h"
#include "cagym_ga3c.h"
#include "cagym_ga3c16.h"
#include "cagym_gen.h"
#include "cagym_dmcts.h"

namespace {

thread_local std::string g_last_error;

struct Env {
    cagym_config cfg;
    CagymDev D;
    std::vector<void*> allocs;
    std::string err;
    bool scenarios_set = false;
    double* sc_obst = nullptr;
    float4* sc_obst_prep = nullptr;
    double* sc_heading_buf = nullptr;
    IgDev G{};
    uint32_t* ig_any = nullptr;
    int32_t* gen_failed = nullptr;  // device word: agents whose rejection loop hit max_tries (cagym_generate_scenarios)
    int32_t* ga3c_ctr = nullptr;    // device words of cagym_ga3c_act's list (k_ga3c_select): ticket, list start, list length
    unsigned char* ga3c_packed = nullptr;   // the weight blob as split f16 operand fragments (k_ga3c_pack16, cagym_ga3c16.h)
    const float* ga3c_packed_src = nullptr; // the blob it was packed from (cagym_ga3c_load_weights)
    int32_t* status_host = nullptr; // host-mapped word the kernels' bounded waits report into (CagymDev::dev_status, cagym_spin.h)
    bool ig_ready = false;
    int any_rvo = 1;
    int obst_rvo = 0;    // RVO agents in worlds with rectangles: the kernels build obstacle half-planes (OBST instantiations)
    int generation = 3;  // CAGYM_KERNEL=v1 selects the one-lane-per-agent kernels (bitwise A/B only)
    int wpw10 = 5;       // worlds per workgroup of the M = 10 kernels (4 while all workgroups are co-resident)
    size_t pre_lds_min = 0;  // CAGYM_PRE_LDS (bytes, read at creation): the PRE half asks for at least this much LDS per workgroup - a cap on how
                             // many of its workgroups share a CU with the caller's policy kernel (tools/cfg4_overlap.py)
    bool begun = false;  // cagym_step_begin was enqueued and no cagym_step_finish has consumed its velocities yet
};

int fail(Env* e, int code, const std::string& msg) {
    g_last_error = msg;
    if (e) e->err = msg;
    return code;
}

#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _s = (call);                                                                      \
        if (_s != hipSuccess)                                                                        \
            return fail(e, CAGYM_E_HIP, std::string(#call) + ": " + hipGetErrorString(_s));          \
    } while (0)

// Every launching entry point runs with the handle's device current (a C caller may hold handles on several devices:
// "one process, 8 handles") and leaves the caller's current device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int dev) {
        status = hipGetDevice(&prev);
        if (status == hipSuccess && prev != dev) {
            status = hipSetDevice(dev);
            switched = status == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define DEVGUARD(e)                                                                                        \
    DeviceGuard _guard((e)->cfg.device);                                                                   \
    if (_guard.status != hipSuccess) return fail(e, CAGYM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_guard.status)); \
    if (int _rc = device_status(e)) return _rc

// A kernel whose bounded intra-workgroup wait expired (cagym_spin.h) wrote its CAGYM_DEVERR_* code into the handle's host-mapped
// status word: the launches since then produced void results.  Every launching entry point refuses to go on (the word is sticky
// until cagym_destroy: the state on the device is not trustworthy any more).
int device_status(Env* e) {
    const int32_t code = e->status_host ? *reinterpret_cast<volatile int32_t*>(e->status_host) : 0;
    if (code == CAGYM_DEVERR_NONE) return CAGYM_OK;
    return fail(e, CAGYM_E_DEVICE, std::string("a kernel of an earlier launch gave up a bounded wait (") +
                                       (code == CAGYM_DEVERR_LP_WAIT ? "LP waves" : code == CAGYM_DEVERR_LASER_WAIT ? "LaserScan passes" : "unknown") +
                                       "): results since then are void, destroy the handle");
}

template <typename T>
int dalloc(Env* e, T** p, size_t n) {
    void* q = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t s = hipMalloc(&q, bytes);
    if (s != hipSuccess) return fail(e, CAGYM_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(s));
    s = hipMemset(q, 0, bytes);
    if (s != hipSuccess) return fail(e, CAGYM_E_HIP, std::string("hipMemset: ") + hipGetErrorString(s));
    e->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return CAGYM_OK;
}

CagymOut to_out(const cagym_outputs* o) {
    CagymOut r{};
    if (o) {
        r.obs_oas = o->obs_oas;
        r.obs_ego = o->obs_ego;
        r.laserscan = o->laserscan;
        r.reward = o->reward;
        r.flags = o->flags;
        r.game_over = o->game_over;
    }
    return r;
}

// worlds per workgroup of the compile-time specialisations (0 = generic: 64 / M worlds, LDS stride 64)
//   M = 4 : 16 worlds,  96 unordered pairs per phase round of 256 lanes
//   M = 10:  5 worlds (225 unordered / 500 directed pair slots, ~35 live agents = 2 rounds of 32 LP groups), or
//            4 worlds while every workgroup of the launch is co-resident (<= 5 per CU): ~28 live agents = one
//            round of LP groups; measured 4096 worlds: 259 vs 247 M env-steps/s, 65536 worlds: 342 vs 392
//   M = 20:  2 worlds, 380 unordered / 800 directed pair slots on 256 lanes (5 workgroups per CU instead of 2)
#define WPW20 2
#define NT20 256  /* lanes per workgroup of the M = 20 specialisation (512: 64 vs 78 M env-steps/s at 2048 x 20) */
// kernel specialisation of the handle: lanes per workgroup, compile-time M (0 = generic) and worlds per workgroup
// (0 = 64 / M worlds, LDS stride 64)
struct Spec2 {
    int nt, mt, wpw;
};
inline Spec2 spec2(const Env* e) {
    const int M = e->cfg.max_agents;
    if (M == 10) return {256, 10, e->wpw10};
    if (M == 4) return {256, 4, 0};
    if (M == 20) return {NT20, 20, WPW20};
    if (M <= 12) return {256, 0, 0};
    return {512, 0, 0};
}
inline int wpw_spec(const Env* e) { return spec2(e).wpw; }
inline int n_wg2(const Env* e) {
    const int M = e->cfg.max_agents;
    const int wpw = wpw_spec(e) ? wpw_spec(e) : CAGYM_WAVE / M;
    return (e->cfg.n_worlds + wpw - 1) / wpw;
}
// obst: the OBST instantiation (worlds may hold rectangles); lines: RVO agents among them (obstacle half-plane rows)
inline size_t lds3_bytes(const Env* e, bool obst, bool lines) {
    const int M = e->cfg.max_agents;
    return cagym_lds3_bytes(M, cagym_as(M, wpw_spec(e)), spec2(e).nt, (obst && lines) ? 2 * e->cfg.max_obstacles : 0, cagym_lpl3(spec2(e).mt, obst), obst, spec2(e).mt);
}
inline bool has_map(const Env* e) { return e->cfg.max_obstacles > 0; }
inline size_t scan_bytes(const Env* e) { return (size_t)e->cfg.n_worlds * e->cfg.max_agents * 16 * sizeof(float); }
inline size_t lds3_bytes(const Env* e) { return lds3_bytes(e, has_map(e), e->obst_rvo != 0); }
// the one-step launch of a handle with rectangles runs on the time-shared layout (carve_lds3_ovl)
inline size_t lds3_step_bytes(const Env* e) {
    if (!has_map(e)) return lds3_bytes(e);
    return cagym_lds3_ovl_bytes(e->cfg.max_agents, cagym_as(e->cfg.max_agents, wpw_spec(e)), spec2(e).nt, e->D.ko);
}
// LP group width of the handle's specialisation (run_steps3)
inline int lp_group_width(const Env* e) {
    const int mt = spec2(e).mt;
    return cagym_gw3(mt);
}

inline int n_waves(const Env* e) {
    int wpw = CAGYM_WAVE / e->cfg.max_agents;
    return (e->cfg.n_worlds + wpw - 1) / wpw;
}

// the one place that maps a handle to its kernel instantiation
// (the launchers live in the per-specialisation translation units, cagym_k3_tu.hip)
struct K3Entry {
    int nt, mt, wp;
    void (*launch[2])(const K3Launch&);  // [OBST]
    void (*setattr[2])(int);
};
#define K3_ROW(NT, MT, WP)                                                                     \
    {NT, MT, WP, {cagym_k3_launch_##NT##_##MT##_##WP##_0, cagym_k3_launch_##NT##_##MT##_##WP##_1}, \
     {cagym_k3_setattr_##NT##_##MT##_##WP##_0, cagym_k3_setattr_##NT##_##MT##_##WP##_1}},
const K3Entry k3_table[] = {CAGYM_K3_SPECS(K3_ROW)};
#undef K3_ROW
inline const K3Entry* k3_entry(const Env* e) {
    const Spec2 sp = spec2(e);
    for (const K3Entry& r : k3_table)
        if (r.nt == sp.nt && r.mt == sp.mt && r.wp == sp.wpw) return &r;
    return nullptr;
}
// one generation-3 launch of the handle's specialisation (free-space or OBST instantiation)
inline void launch3(const Env* e, bool rollout, bool auto_reset, const float* ext, int n_steps, const CagymOut& o, hipStream_t st) {
    K3Launch L;
    L.D = e->D; L.ext = ext; L.out = o; L.n_steps = n_steps; L.any_rvo = e->any_rvo; L.rollout = rollout; L.auto_reset = auto_reset;
    L.grid = (unsigned)n_wg2(e); L.lds = rollout ? lds3_bytes(e) : lds3_step_bytes(e); L.stream = st;
#ifdef CAGYM_DIAG_LDS_PAD  // occupancy experiments only (tools/README.md): unused LDS bytes on top, to force fewer workgroups per CU
    if (const char* pad = getenv("CAGYM_LDS_PAD")) L.lds += (size_t)atoi(pad);
#endif
    k3_entry(e)->launch[has_map(e) ? 1 : 0](L);
}

}  // namespace

extern "C" {

int cagym_version(void) { return CAGYM_VERSION; }

const char* cagym_last_error(void* env) {
    Env* e = reinterpret_cast<Env*>(env);
    return e ? e->err.c_str() : g_last_error.c_str();
}

int cagym_create(const cagym_config* cfg, void** env_out) {
    if (!cfg || !env_out) return fail(nullptr, CAGYM_E_INVALID, "cagym_create: null argument");
    *env_out = nullptr;
    if (cfg->n_worlds < 1) return fail(nullptr, CAGYM_E_INVALID, "n_worlds must be >= 1");
    if (cfg->max_agents < 2 || cfg->max_agents > 32)
        return fail(nullptr, CAGYM_E_UNSUPPORTED, "max_agents must be in [2, 32] (a world may not straddle a wavefront)");
    if (cfg->n_scenarios < cfg->n_worlds) return fail(nullptr, CAGYM_E_INVALID, "n_scenarios must be >= n_worlds");
    if ((long long)cfg->n_worlds * cfg->max_agents >= (1ll << 31) || (long long)cfg->n_scenarios * cfg->max_agents >= (1ll << 31))
        return fail(nullptr, CAGYM_E_UNSUPPORTED, "n_worlds * max_agents (and n_scenarios * max_agents) must stay below 2^31: the kernels keep flat agent indices in 32 bits");
    if (cfg->max_obstacles < 0 || !(cfg->dt > 0)) return fail(nullptr, CAGYM_E_INVALID, "bad max_obstacles / dt");
    if (cfg->rvo_max_neighbors < 0) return fail(nullptr, CAGYM_E_INVALID, "rvo_max_neighbors must be >= 0 (0 = max_agents)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, CAGYM_E_NODEVICE, "no HIP device: libcagym_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, CAGYM_E_INVALID, "device ordinal out of range");
    Env* e = new Env();
    e->cfg = *cfg;
    DeviceGuard guard(cfg->device);
    if (guard.status != hipSuccess) {
        delete e;
        return fail(nullptr, CAGYM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.status));
    }
    CagymDev& D = e->D;
    memset(&D, 0, sizeof(D));
    const size_t N = cfg->n_worlds, M = cfg->max_agents, S = cfg->n_scenarios, NM = N * M, SM = S * M;
    D.N = (int)N; D.M = (int)M; D.S = (int)S; D.Kobs = cfg->max_obstacles;
    D.go_mode = cfg->game_over_mode; D.collide_static = cfg->collide_with_static; D.laserscan = cfg->laserscan;
    D.dt = cfg->dt;
    D.inv_dt = 1.0 / cfg->dt;
    D.maxnb = cfg->rvo_max_neighbors > 0 ? cfg->rvo_max_neighbors : (int)M;  // RVOPolicy.py:15: Config.MAX_NUM_AGENTS_IN_ENVIRONMENT
    if (D.maxnb > (int)M - 1) D.maxnb = (int)M - 1;                            // there are at most M - 1 other agents
    int rc = CAGYM_OK;
    double* d6 = nullptr; double* dcoop = nullptr;
    int32_t *dpol = nullptr, *ddyn = nullptr, *dna = nullptr, *dno = nullptr;
    uint32_t* dmap = nullptr;
#define A(call) if ((rc = (call)) != CAGYM_OK) { cagym_destroy(e); return rc; }
    A(dalloc(e, &d6, SM * 6)); A(dalloc(e, &e->sc_heading_buf, SM)); A(dalloc(e, &dcoop, SM));
    A(dalloc(e, &dpol, SM)); A(dalloc(e, &ddyn, SM)); A(dalloc(e, &dna, S)); A(dalloc(e, &dno, S));
    if (cfg->max_obstacles > 0) {
        A(dalloc(e, &dmap, S * CAGYM_MAPD * CAGYM_MAPW));
        A(dalloc(e, &e->sc_obst, S * (size_t)cfg->max_obstacles * 4));
        A(dalloc(e, &e->sc_obst_prep, S * (size_t)cfg->max_obstacles * 4));
    }
    D.sc_agents6 = d6; D.sc_heading0 = nullptr; D.sc_coop = dcoop; D.sc_policy = dpol; D.sc_dyn = ddyn;
    D.sc_nagents = dna; D.sc_nobst = dno; D.map_bits = dmap; D.sc_obst = e->sc_obst; D.sc_obst_prep = e->sc_obst_prep;
    A(dalloc(e, &D.px, NM)); A(dalloc(e, &D.py, NM)); A(dalloc(e, &D.vx, NM)); A(dalloc(e, &D.vy, NM));
    A(dalloc(e, &D.heading, NM)); A(dalloc(e, &D.heading_ego, NM)); A(dalloc(e, &D.dist_goal, NM));
    A(dalloc(e, &D.time_rem, NM)); A(dalloc(e, &D.t, NM)); A(dalloc(e, &D.gx, NM)); A(dalloc(e, &D.gy, NM));
    A(dalloc(e, &D.radius, NM)); A(dalloc(e, &D.pref, NM)); A(dalloc(e, &D.speed, NM)); A(dalloc(e, &D.dhead, NM));
    A(dalloc(e, &D.aux0, NM)); A(dalloc(e, &D.aux1, NM)); A(dalloc(e, &D.coop, NM));
    A(dalloc(e, &D.action, NM * 2)); A(dalloc(e, &D.status, NM)); A(dalloc(e, &D.step_num, NM));
    A(dalloc(e, &D.n_observed, NM));
    A(dalloc(e, &D.lp_vel, NM));
    A(dalloc(e, &D.n_agents, N)); A(dalloc(e, &D.episode, N)); A(dalloc(e, &D.ep_len, N));
    A(dalloc(e, &D.ep_return, N)); A(dalloc(e, &D.stat_return, N)); A(dalloc(e, &D.stat_episodes, N));
    A(dalloc(e, &D.stat_steps, N)); A(dalloc(e, &D.stat_outcomes, N * 3));
    A(dalloc(e, &e->ga3c_ctr, 4));  // at creation: cagym_ga3c_act may run inside a stream capture (no allocation there)
    A(dalloc(e, &e->ga3c_packed, GA16_PACKED_BYTES));
#undef A
    {   // the kernels' status word: pinned host memory mapped into the device's address space (written only when a bounded wait expires)
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            cagym_destroy(e);
            return fail(nullptr, CAGYM_E_NOMEM, "hipHostMalloc of the device status word failed");
        }
        memset(hp, 0, 64);
        e->status_host = reinterpret_cast<int32_t*>(hp);
        D.dev_status = reinterpret_cast<int32_t*>(dp);
    }
    e->err.clear();
    size_t lds = cagym_lds_bytes((int)M);
    if (lds > 160 * 1024) { cagym_destroy(e); return fail(nullptr, CAGYM_E_UNSUPPORTED, "LDS budget exceeded"); }
    // > 64 KiB of dynamic LDS needs the attribute raised
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_reset), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    {
        const char* g = getenv("CAGYM_KERNEL");
        if (g && (!strcmp(g, "v1") || !strcmp(g, "1"))) e->generation = 1;
        else if (g && g[0] && strcmp(g, "v3") && strcmp(g, "3")) {  // a retired or misspelt generation must not silently run the default
            cagym_destroy(e);
            return fail(nullptr, CAGYM_E_INVALID, std::string("CAGYM_KERNEL=") + g + ": unknown kernel generation (v1 or v3)");
        }
        {
            hipDeviceProp_t prop;
            int cus = 256;
            if (hipGetDeviceProperties(&prop, e->cfg.device) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
            e->wpw10 = (e->cfg.n_worlds + 3) / 4 <= 4 * cus ? 4 : 5;  // 4 workgroups per CU fit (128 VGPRs; 30.9 KB of LDS with 4 worlds, 37.7 KB with 5)
            if (has_map(e) && e->wpw10 == 5) {
                // worlds with rectangles (OBST instantiation, 135 VGPRs: at most 3 workgroups per CU): 4 worlds per workgroup when their
                // smaller LDS footprint buys a workgroup per CU (cfg4: 53.4 KB -> 3 per CU, 65.9 KB with 5 worlds -> 2;
                // profiles/r3/cfg4_occupancy_ab.txt)
                auto per_cu = [&](int wpw) {
                    e->wpw10 = wpw;
                    size_t b = lds3_bytes(e, true, true);
                    if (b > 160 * 1024) b = lds3_bytes(e, true, false);
                    const int n = (int)((size_t)160 * 1024 / b);
                    return n < 3 ? n : 3;
                };
                const int n4 = per_cu(4), n5 = per_cu(5);
                e->wpw10 = n4 > n5 ? 4 : 5;
            }
            if (const char* pl = getenv("CAGYM_PRE_LDS")) e->pre_lds_min = (size_t)atol(pl);
            const char* w = getenv("CAGYM_WPW10");  // diagnostics
            if (w && (w[0] == '4' || w[0] == '5')) e->wpw10 = w[0] - '0';
        }
        int lds3 = (int)lds3_bytes(e, false, false), lds3_obst = (int)lds3_bytes(e, true, true);
        if (lds3_obst > 160 * 1024) lds3_obst = (int)lds3_bytes(e, true, false);  // too many rectangles for RVO agents: refused at set_scenarios
        if (e->generation == 3 && lds3 > 160 * 1024) e->generation = 1;
        // the free-space kernels keep the neighbour keys in the LP scratch (cagym_dsq_aliased): it must hold them
        if (e->generation == 3 && cagym_dsq_aliased(false, spec2(e).mt) &&
            (size_t)cagym_as((int)M, wpw_spec(e)) * cagym_mp((int)M) * 8 > (size_t)cagym_lpl3(spec2(e).mt, false) * spec2(e).nt * 16) {
            cagym_destroy(e);
            return fail(nullptr, CAGYM_E_UNSUPPORTED, "neighbour keys do not fit the LP scratch of this specialisation");
        }
        if (!k3_entry(e)) { cagym_destroy(e); return fail(nullptr, CAGYM_E_UNSUPPORTED, "no kernel specialisation for this shape"); }
#ifdef CAGYM_DIAG_LDS_PAD
        if (const char* pad = getenv("CAGYM_LDS_PAD")) { lds3 += atoi(pad); lds3_obst += atoi(pad); }
#endif
        k3_entry(e)->setattr[0](lds3);
        if (lds3_obst <= 160 * 1024) k3_entry(e)->setattr[1](lds3_obst);
    }
    (void)hipGetLastError();
    *env_out = e;
    return CAGYM_OK;
}

int cagym_destroy(void* env) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return CAGYM_OK;
    for (void* p : e->allocs)
        if (p) (void)hipFree(p);
    if (e->status_host) (void)hipHostFree(e->status_host);
    delete e;
    return CAGYM_OK;
}

int cagym_set_scenarios(void* env, const double* agents6, const double* heading0, const int32_t* policy_id,
                        const int32_t* dynamics_id, const int32_t* n_agents, const double* coop,
                        const double* obstacles, const int32_t* n_obst, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!agents6 || !policy_id || !dynamics_id) return fail(e, CAGYM_E_INVALID, "agents6 / policy_id / dynamics_id are required");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DEVGUARD(e);
    const size_t S = e->cfg.n_scenarios, M = e->cfg.max_agents, SM = S * M;
    // validate ids on the host: the kernels index switch tables with them
    for (size_t k = 0; k < SM; k++) {
        if (policy_id[k] < 0 || policy_id[k] > CAGYM_POL_IGMCTS) return fail(e, CAGYM_E_INVALID, "policy id out of range");
        if (dynamics_id[k] < 0 || dynamics_id[k] > CAGYM_DYN_FIRSTORDER) return fail(e, CAGYM_E_INVALID, "dynamics id out of range");
    }
    // decided on locals, committed only after every validation and copy below succeeded: a refused call leaves the handle as it was
    int new_any_rvo = 0, new_obst_rvo = 0, new_ko = 0;
    for (size_t k = 0; k < SM; k++)
        if (policy_id[k] == CAGYM_POL_RVO) new_any_rvo = 1;
    std::vector<int32_t> na(S), no(S, 0);
    for (size_t s = 0; s < S; s++) {
        na[s] = n_agents ? n_agents[s] : (int32_t)M;
        if (na[s] < 0 || na[s] > (int32_t)M) return fail(e, CAGYM_E_INVALID, "n_agents out of range");
        if (n_obst && e->cfg.max_obstacles > 0) {
            no[s] = n_obst[s];
            if (no[s] < 0 || no[s] > e->cfg.max_obstacles) return fail(e, CAGYM_E_INVALID, "n_obst out of range");
        }
    }
    // RVO agents among rectangles: the kernels build obstacle half-planes (RVOPolicy.py:56-57).  Capacity of an LP group:
    // 4 half-planes per lane; an agent sees at most 2 edges of a rectangle from their right side.
    {
        bool any_obst = false;
        for (size_t sc = 0; sc < S; sc++) any_obst |= no[sc] > 0;
        if (any_obst && !obstacles) return fail(e, CAGYM_E_INVALID, "n_obst > 0 needs the obstacles array");
        new_obst_rvo = (any_obst && new_any_rvo) ? 1 : 0;
        new_ko = new_obst_rvo ? 2 * e->cfg.max_obstacles : 0;
        if (new_obst_rvo) {
            const int K = e->cfg.max_obstacles, gw = lp_group_width(e);
            const Spec2 sp = spec2(e);
            const int as = cagym_as((int)M, sp.wpw);
            if (e->generation != 3) return fail(e, CAGYM_E_UNSUPPORTED, "RVO agents among obstacles need the generation-3 kernels");
            // LP group capacity; coverage bit masks (<= 32 obstacle lines per ego); the obstacle-neighbour lists (8 B per candidate + 4 B per
            // work item + 1 B per rank) borrow the LP3 scratch
            if (2 * K + (int)M - 1 > 4 * gw || 2 * K > 32 || (size_t)2 * K * as * 13 > (size_t)4 * sp.nt * 16 || lds3_bytes(e, true, true) > 160 * 1024)
                return fail(e, CAGYM_E_UNSUPPORTED, "too many rectangles per world for RVO agents at this max_agents (2 * max_obstacles + max_agents - 1 half-planes per ego)");
            for (size_t sc = 0; sc < S; sc++)
                    for (int k = 0; k < no[sc]; k++) {
                        const double* r = obstacles + (sc * K + k) * 4;
                        if (!(r[2] > r[0]) || !(r[3] > r[1]))
                            return fail(e, CAGYM_E_INVALID, "RVO agents need non-degenerate rectangles (xl < xu, yl < yu)");
                    }
        }
    }
    std::vector<double> cp(SM, 1.0);  // agent.py:10
    if (coop) memcpy(cp.data(), coop, SM * sizeof(double));
    CagymDev& D = e->D;
    double* dh0 = e->sc_heading_buf;
    HIPCHK(e, hipMemcpyAsync(const_cast<double*>(D.sc_agents6), agents6, SM * 6 * sizeof(double), hipMemcpyHostToDevice, st));
    if (heading0) {
        HIPCHK(e, hipMemcpyAsync(dh0, heading0, SM * sizeof(double), hipMemcpyHostToDevice, st));
        D.sc_heading0 = dh0;
    } else {
        D.sc_heading0 = nullptr;
    }
    HIPCHK(e, hipMemcpyAsync(const_cast<int32_t*>(D.sc_policy), policy_id, SM * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(const_cast<int32_t*>(D.sc_dyn), dynamics_id, SM * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(const_cast<int32_t*>(D.sc_nagents), na.data(), S * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(const_cast<double*>(D.sc_coop), cp.data(), SM * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(const_cast<int32_t*>(D.sc_nobst), no.data(), S * sizeof(int32_t), hipMemcpyHostToDevice, st));
    std::vector<float> prep;  // staged until the stream synchronisation below
    if (e->cfg.max_obstacles > 0) {
        if (obstacles) {
            HIPCHK(e, hipMemcpyAsync(e->sc_obst, obstacles, S * (size_t)e->cfg.max_obstacles * 4 * sizeof(double), hipMemcpyHostToDevice, st));
            // RVOSimulator::addObstacle per rectangle (vertices narrowed to float as Cython does): unit directions
            // normalize(next - this) = v * (1.0f / |v|) and convexity leftOf(prev, this, next) >= 0, in plain IEEE fp32
            const size_t K = (size_t)e->cfg.max_obstacles;
            prep.assign(S * K * 16, 0.0f);
            for (size_t r = 0; r < S * K; r++) {
                const float xl = (float)obstacles[4 * r], yl = (float)obstacles[4 * r + 1], xu = (float)obstacles[4 * r + 2], yu = (float)obstacles[4 * r + 3];
                const float X[4] = {xu, xl, xl, xu}, Y[4] = {yu, yu, yl, yl};
                float* q = prep.data() + 16 * r;
                q[0] = xl; q[1] = yl; q[2] = xu; q[3] = yu;
                uint32_t convex = 0;
                for (int k = 0; k < 4; k++) {
                    const int nx = (k + 1) & 3, pv = (k + 3) & 3;
                    const volatile float ex = X[nx] - X[k], ey = Y[nx] - Y[k];
                    const volatile float sq = ex * ex;
                    const volatile float sq2 = ey * ey;
                    const volatile float len2 = sq + sq2;
                    const volatile float inv = 1.0f / sqrtf(len2);
                    q[4 + 2 * k] = ex * inv;
                    q[5 + 2 * k] = ey * inv;
                    const volatile float a0 = X[pv] - X[nx], a1 = Y[pv] - Y[nx], b0 = X[k] - X[pv], b1 = Y[k] - Y[pv];
                    const volatile float m0 = a0 * b1;
                    const volatile float m1 = a1 * b0;
                    if (m0 - m1 >= 0.0f) convex |= 1u << k;
                }
                memcpy(&q[12], &convex, 4);
                // its raster footprint needs neither numpy's negative-index wrap nor clamping (laser_chunk3's shortcut)
                q[13] = (xl > -14.7f && yl > -14.7f && xu < 14.7f && yu < 14.7f) ? 1.0f : 0.0f;
            }
            HIPCHK(e, hipMemcpyAsync(e->sc_obst_prep, prep.data(), prep.size() * sizeof(float), hipMemcpyHostToDevice, st));
        }
        hipLaunchKernelGGL(k_rasterize, dim3((unsigned)S), dim3(256), 0, st, e->sc_obst, D.sc_nobst, e->cfg.max_obstacles,
                           const_cast<uint32_t*>(D.map_bits));
        HIPCHK(e, hipGetLastError());
    }
    // host staging vectors die at return: the copies above must have consumed them
    HIPCHK(e, hipStreamSynchronize(st));
    // a new pool restarts the episode numbering
    HIPCHK(e, hipMemsetAsync(D.episode, 0, e->cfg.n_worlds * sizeof(int32_t), st));
    e->any_rvo = new_any_rvo;
    e->obst_rvo = new_obst_rvo;
    e->D.ko = new_ko;
    e->scenarios_set = true;
    return CAGYM_OK;
}

int cagym_generate_scenarios(void* env, const cagym_gen_params* params, int32_t* n_failed_host, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!params) return fail(e, CAGYM_E_INVALID, "null params");
    const cagym_gen_params& P = *params;
    const int M = e->cfg.max_agents;
    if (P.n_min < 1 || P.n_max > M || P.n_min > P.n_max) return fail(e, CAGYM_E_INVALID, "need 1 <= n_min <= n_max <= max_agents");
    const int32_t pols[3] = {P.ego_policy, P.policy_a, P.policy_b};
    for (int32_t q : pols)
        if (q < 0 || q > CAGYM_POL_IGMCTS) return fail(e, CAGYM_E_INVALID, "policy id out of range");
    if (P.ego_dynamics < 0 || P.ego_dynamics > CAGYM_DYN_FIRSTORDER || P.other_dynamics < 0 || P.other_dynamics > CAGYM_DYN_FIRSTORDER)
        return fail(e, CAGYM_E_INVALID, "dynamics id out of range");
    if (P.max_tries < 1 || !(P.side > 0) || !(P.p_b >= 0 && P.p_b <= 1)) return fail(e, CAGYM_E_INVALID, "bad generator parameters");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DEVGUARD(e);
    CagymDev& D = e->D;
    GenDev G;
    G.agents6 = const_cast<double*>(D.sc_agents6);
    G.policy = const_cast<int32_t*>(D.sc_policy);
    G.dyn = const_cast<int32_t*>(D.sc_dyn);
    G.nagents = const_cast<int32_t*>(D.sc_nagents);
    G.coop = const_cast<double*>(D.sc_coop);
    G.nobst = const_cast<int32_t*>(D.sc_nobst);
    G.S = e->cfg.n_scenarios;
    G.M = M;
    if (!e->gen_failed) {
        int rcf = dalloc(e, &e->gen_failed, 1);
        if (rcf != CAGYM_OK) return rcf;
    }
    int32_t* d_failed = e->gen_failed;
    HIPCHK(e, hipMemsetAsync(d_failed, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_generate_scenarios, dim3((G.S + 63) / 64), dim3(64), 0, st, G, P, d_failed);
    HIPCHK(e, hipGetLastError());
    D.sc_heading0 = nullptr;  // toward the goal (agent.py:29-31)
    e->any_rvo = (P.ego_policy == CAGYM_POL_RVO || P.policy_a == CAGYM_POL_RVO || P.policy_b == CAGYM_POL_RVO) ? 1 : 0;
    e->obst_rvo = 0;  // the generator draws free-space worlds
    e->D.ko = 0;
    if (e->cfg.max_obstacles > 0) {  // free space: empty rasters
        hipLaunchKernelGGL(k_rasterize, dim3((unsigned)G.S), dim3(256), 0, st, e->sc_obst, D.sc_nobst, e->cfg.max_obstacles,
                           const_cast<uint32_t*>(D.map_bits));
        HIPCHK(e, hipGetLastError());
    }
    if (n_failed_host) {
        HIPCHK(e, hipMemcpyAsync(n_failed_host, d_failed, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipStreamSynchronize(st));
    }
    HIPCHK(e, hipMemsetAsync(D.episode, 0, e->cfg.n_worlds * sizeof(int32_t), st));  // a new pool restarts the episode numbering
    e->scenarios_set = true;
    return CAGYM_OK;
}

int cagym_get_scenarios(void* env, cagym_scenario_ptrs* out) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!out) return fail(e, CAGYM_E_INVALID, "null out");
    out->agents6 = e->D.sc_agents6;
    out->policy = e->D.sc_policy;
    out->dynamics = e->D.sc_dyn;
    out->n_agents = e->D.sc_nagents;
    out->coop = e->D.sc_coop;
    return CAGYM_OK;
}

int cagym_reset(void* env, const uint8_t* world_mask, int advance_episode, const cagym_outputs* out, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_reset before cagym_set_scenarios");
    DEVGUARD(e);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CagymOut o = to_out(out);
    e->begun = false;
    hipLaunchKernelGGL(k_reset, dim3(n_waves(e)), dim3(64), cagym_lds_bytes(e->cfg.max_agents), st, e->D, world_mask,
                       advance_episode, o);
    HIPCHK(e, hipGetLastError());
    if (e->cfg.laserscan && o.laserscan) return cagym_laserscan(env, o.laserscan, stream);
    return CAGYM_OK;
}

int cagym_step(void* env, const float* ext_actions, const cagym_outputs* out, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_step before cagym_set_scenarios");
    DEVGUARD(e);
    e->begun = false;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CagymOut o = to_out(out);
    if (!e->cfg.laserscan) o.laserscan = nullptr;
    if (e->generation == 3) {
        launch3(e, false, false, ext_actions, 1, o, st);
    } else
    hipLaunchKernelGGL(k_step, dim3(n_waves(e)), dim3(64), cagym_lds_bytes(e->cfg.max_agents), st, e->D, ext_actions, o);
    HIPCHK(e, hipGetLastError());
    if (e->generation != 3 && e->cfg.laserscan && o.laserscan) return cagym_laserscan(env, o.laserscan, stream);  // generation 3 scans in-kernel
    // ... in its OBST instantiation; a handle without rectangles runs the free-space kernels: every beam of an empty map reads 0.0
    // (LaserScanSensor.py:27-58 on an all-free Map)
    if (e->generation == 3 && !has_map(e) && o.laserscan) HIPCHK(e, hipMemsetAsync(o.laserscan, 0, scan_bytes(e), st));
    return CAGYM_OK;
}

int cagym_step_autoreset(void* env, const float* ext_actions, const cagym_outputs* out, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_step_autoreset before cagym_set_scenarios");
    DEVGUARD(e);
    e->begun = false;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CagymOut o = to_out(out);
    if (!e->cfg.laserscan) o.laserscan = nullptr;
    if (e->generation == 3) {
        launch3(e, false, true, ext_actions, 1, o, st);
    } else
        return fail(e, CAGYM_E_UNSUPPORTED, "cagym_step_autoreset needs the generation-3 kernels");
    HIPCHK(e, hipGetLastError());
    if (!has_map(e) && o.laserscan) HIPCHK(e, hipMemsetAsync(o.laserscan, 0, scan_bytes(e), st));  // empty map: 0.0 everywhere
    return CAGYM_OK;  // the scan of the (possibly restarted) worlds is part of the launch
}

// ---- the split step (csrc/cagym_split3.h) -----------------------------------------------------------------------------------------
int cagym_step_begin(void* env, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_step_begin before cagym_set_scenarios");
    if (e->generation != 3) return fail(e, CAGYM_E_UNSUPPORTED, "the split step needs the generation-3 kernels");
    DEVGUARD(e);
    e->begun = true;
    if (!e->any_rvo) return CAGYM_OK;  // no internal RVO policy: nothing to solve ahead of the external actions
    K3Launch L;
    L.D = e->D; L.ext = nullptr; L.out = CagymOut{}; L.n_steps = 1; L.any_rvo = 1; L.rollout = false; L.auto_reset = false;
    L.half = K3_HALF_PRE;
    L.grid = (unsigned)n_wg2(e);
    L.lds = cagym_lds3_pre_bytes(e->cfg.max_agents, cagym_as(e->cfg.max_agents, wpw_spec(e)), spec2(e).nt, has_map(e) ? e->D.ko : 0);
    if (L.lds < e->pre_lds_min && e->pre_lds_min <= 64 * 1024) L.lds = e->pre_lds_min;
    L.stream = reinterpret_cast<hipStream_t>(stream);
    k3_entry(e)->launch[has_map(e) ? 1 : 0](L);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_step_finish(void* env, const float* ext_actions, const cagym_outputs* out, int auto_reset, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_step_finish before cagym_set_scenarios");
    if (e->generation != 3) return fail(e, CAGYM_E_UNSUPPORTED, "the split step needs the generation-3 kernels");
    if (!e->begun) return fail(e, CAGYM_E_STATE, "cagym_step_finish without a cagym_step_begin on the current state");
    DEVGUARD(e);
    e->begun = false;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CagymOut o = to_out(out);
    if (!e->cfg.laserscan) o.laserscan = nullptr;
    K3Launch L;
    L.D = e->D; L.ext = ext_actions; L.out = o; L.n_steps = 1; L.any_rvo = 0; L.rollout = false; L.auto_reset = auto_reset != 0;
    L.half = K3_HALF_POST;
    L.grid = (unsigned)n_wg2(e);
    L.lds = cagym_lds3_post_bytes(e->cfg.max_agents, cagym_as(e->cfg.max_agents, wpw_spec(e)), has_map(e) ? e->D.ko / 2 : 0, has_map(e));
    L.stream = st;
    k3_entry(e)->launch[has_map(e) ? 1 : 0](L);
    HIPCHK(e, hipGetLastError());
    if (!has_map(e) && o.laserscan) HIPCHK(e, hipMemsetAsync(o.laserscan, 0, scan_bytes(e), st));  // empty map: 0.0 everywhere
    return CAGYM_OK;
}

int cagym_rollout(void* env, int n_steps, int auto_reset, const cagym_outputs* out, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_rollout before cagym_set_scenarios");
    DEVGUARD(e);
    e->begun = false;
    if (n_steps < 1) return fail(e, CAGYM_E_INVALID, "n_steps must be >= 1");
    if (e->cfg.laserscan && out && out->laserscan && e->generation != 3)
        return fail(e, CAGYM_E_UNSUPPORTED, "cagym_rollout produces laserscan with the generation-3 kernels only");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CagymOut o = to_out(out);
    if (!e->cfg.laserscan) o.laserscan = nullptr;
    size_t lds = cagym_lds_bytes(e->cfg.max_agents);
    if (e->generation == 3) {
        launch3(e, true, auto_reset != 0, nullptr, n_steps, o, st);
    } else if (auto_reset)
        hipLaunchKernelGGL(k_rollout<true>, dim3(n_waves(e)), dim3(64), lds, st, e->D, n_steps, o);
    else
        hipLaunchKernelGGL(k_rollout<false>, dim3(n_waves(e)), dim3(64), lds, st, e->D, n_steps, o);
    HIPCHK(e, hipGetLastError());
    if (e->generation == 3 && !has_map(e) && o.laserscan)  // empty map: 0.0 everywhere, all n_steps slices
        HIPCHK(e, hipMemsetAsync(o.laserscan, 0, (size_t)n_steps * scan_bytes(e), st));
    return CAGYM_OK;
}

int cagym_kernel_name(void* env, int rollout, int auto_reset, char* buf, int buf_len) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e || !buf || buf_len < 1) return fail(e, CAGYM_E_INVALID, "bad arguments");
    if (e->generation == 3) {
        const Spec2 sp = spec2(e);
        snprintf(buf, (size_t)buf_len, "%s%d<%d, %d, %d, %s, %s>", rollout ? "k_rollout" : "k_step", e->generation, sp.nt, sp.mt,
                 sp.wpw, auto_reset ? "true" : "false", has_map(e) ? "true" : "false");
    } else if (rollout) {
        snprintf(buf, (size_t)buf_len, "k_rollout<%s>", auto_reset ? "true" : "false");
    } else {
        snprintf(buf, (size_t)buf_len, "k_step");
    }
    return CAGYM_OK;
}

int cagym_laserscan(void* env, float* laserscan, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!laserscan) return fail(e, CAGYM_E_INVALID, "null laserscan buffer");
    DEVGUARD(e);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    size_t total = (size_t)e->cfg.n_worlds * e->cfg.max_agents * 16;
    hipLaunchKernelGGL(k_laserscan, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, e->D, laserscan);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_occupancy_grid(void* env, uint8_t* grid, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!grid) return fail(e, CAGYM_E_INVALID, "null grid buffer");
    if (e->cfg.max_obstacles <= 0) return fail(e, CAGYM_E_STATE, "cagym_occupancy_grid needs an env created with max_obstacles > 0");
    DEVGUARD(e);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_occupancy_grid, dim3((unsigned)((size_t)e->cfg.n_worlds * e->cfg.max_agents)), dim3(256), 0, st, e->D, grid);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_get_state(void* env, cagym_state_ptrs* out) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e || !out) return fail(e, CAGYM_E_INVALID, "null argument");
    const CagymDev& D = e->D;
    out->pos_x = D.px; out->pos_y = D.py; out->vel_x = D.vx; out->vel_y = D.vy; out->heading = D.heading;
    out->heading_ego = D.heading_ego; out->dist_to_goal = D.dist_goal; out->time_remaining = D.time_rem; out->t = D.t;
    out->goal_x = D.gx; out->goal_y = D.gy; out->radius = D.radius; out->pref_speed = D.pref; out->speed = D.speed;
    out->delta_heading = D.dhead; out->aux0 = D.aux0; out->aux1 = D.aux1;
    out->action = D.action; out->status = D.status; out->step_num = D.step_num; out->n_agents = D.n_agents;
    out->n_observed = D.n_observed; out->episode = D.episode; out->map_bits = const_cast<uint32_t*>(D.map_bits);
    out->stat_return = D.stat_return; out->stat_episodes = D.stat_episodes; out->stat_steps = D.stat_steps;
    out->stat_outcomes = D.stat_outcomes;
    return CAGYM_OK;
}

// one 24-byte record per world (cagym.h: cagym_pack_episode_stats)
__global__ void __launch_bounds__(256) k_pack_stats(CagymDev D, int32_t* __restrict__ rec) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= D.N) return;
    int2* r = reinterpret_cast<int2*>(rec + (size_t)w * 6);  // 24-byte records: 8-byte aligned
    r[0] = make_int2(__float_as_int(D.stat_return[w]), D.stat_episodes[w]);
    r[1] = make_int2(D.stat_steps[w], D.stat_outcomes[3 * w]);
    r[2] = make_int2(D.stat_outcomes[3 * w + 1], D.stat_outcomes[3 * w + 2]);
}

int cagym_pack_episode_stats(void* env, int32_t* records, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!records) return fail(e, CAGYM_E_INVALID, "null records buffer");
    DEVGUARD(e);
    hipLaunchKernelGGL(k_pack_stats, dim3((unsigned)((e->cfg.n_worlds + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->D, records);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

// which forward kernel (read per call: the A/B tests flip it inside one process): default = split-f16 matrix cores
// (cagym_ga3c16.h); CAGYM_GA3C=mfma32: round 2's exact-fp32 matrix-core kernel; =valu: round 1's vector kernel
enum { GA_KERNEL_H16 = 0, GA_KERNEL_MFMA32 = 1, GA_KERNEL_VALU = 2, GA_KERNEL_BAD = -1 };
static int ga3c_kernel_choice() {
    const char* which = getenv("CAGYM_GA3C");
    if (!which || !which[0] || !strcmp(which, "h16")) return GA_KERNEL_H16;
    if (!strcmp(which, "mfma32")) return GA_KERNEL_MFMA32;
    if (!strcmp(which, "valu")) return GA_KERNEL_VALU;
    return GA_KERNEL_BAD;
}

// the handle's packed copy of `weights`: made on `st` the first time a blob (by address) is used; a caller that rewrites the
// blob in place says so with cagym_ga3c_load_weights
static int ga3c_pack(Env* e, const float* weights, hipStream_t st, bool force) {
    if (!force && e->ga3c_packed_src == weights) return CAGYM_OK;
    hipLaunchKernelGGL(k_ga3c_pack16, dim3((GA16_PACK_THREADS + 255) / 256), dim3(256), 0, st, weights, e->ga3c_packed);
    const hipError_t s = hipGetLastError();
    if (s != hipSuccess) {
        e->ga3c_packed_src = nullptr;
        return fail(e, CAGYM_E_HIP, std::string("k_ga3c_pack16: ") + hipGetErrorString(s));
    }
    e->ga3c_packed_src = weights;
    return CAGYM_OK;
}

int cagym_ga3c_load_weights(void* env, const float* weights, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!weights) return fail(e, CAGYM_E_INVALID, "null weights");
    DEVGUARD(e);
    return ga3c_pack(e, weights, reinterpret_cast<hipStream_t>(stream), true);
}

static void launch_ga3c_state(Env* e, int max_observed, const int32_t* agent_idx, long long rows, uint32_t* ctr, float* state,
                              hipStream_t st) {
    if (e->cfg.max_agents <= 16)
        hipLaunchKernelGGL(k_ga3c_state<16>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, e->D, max_observed, agent_idx, (int)rows, ctr, state);
    else
        hipLaunchKernelGGL(k_ga3c_state<32>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, e->D, max_observed, agent_idx, (int)rows, ctr, state);
}

int cagym_ga3c_state(void* env, int max_observed, float* state, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!state || max_observed < 1 || max_observed > 10) return fail(e, CAGYM_E_INVALID, "bad arguments (max_observed in 1..10)");
    DEVGUARD(e);
    launch_ga3c_state(e, max_observed, nullptr, (long long)e->cfg.n_worlds * e->cfg.max_agents, nullptr, state, reinterpret_cast<hipStream_t>(stream));
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

size_t cagym_ga3c_act_workspace_bytes(void* env) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return 0;
    const size_t total = (size_t)e->cfg.n_worlds * e->cfg.max_agents;
    return 256 + a16(total * sizeof(int32_t)) + total * 76 * sizeof(float);  // [count | agent list | state rows]
}

int cagym_ga3c_act(void* env, const float* weights, int max_observed, void* work, float* ext_actions, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!weights || !work || !ext_actions || max_observed < 1 || max_observed > 10)
        return fail(e, CAGYM_E_INVALID, "bad arguments (max_observed in 1..10)");
    DEVGUARD(e);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int which = ga3c_kernel_choice();
    if (which == GA_KERNEL_BAD) return fail(e, CAGYM_E_INVALID, "CAGYM_GA3C: unknown forward kernel (h16, mfma32 or valu)");
    if (which == GA_KERNEL_H16) {  // one launch: list, state rows and network per workgroup of 32 worlds (cagym_ga3c16.h); `work` is not used
        if (int rc = ga3c_pack(e, weights, st, false)) return rc;
        const unsigned grid = (unsigned)((e->cfg.n_worlds + 31) / 32);
        if (e->cfg.max_agents <= 16)
            hipLaunchKernelGGL(k_ga3c_act_h16<16>, dim3(grid), dim3(512), 0, st, e->D, e->ga3c_packed, max_observed, ext_actions);
        else
            hipLaunchKernelGGL(k_ga3c_act_h16<32>, dim3(grid), dim3(512), 0, st, e->D, e->ga3c_packed, max_observed, ext_actions);
        HIPCHK(e, hipGetLastError());
        return CAGYM_OK;
    }
    // CAGYM_GA3C=mfma32 / valu (A/B): the three-launch chain of rounds 2 - 3
    const size_t total = (size_t)e->cfg.n_worlds * e->cfg.max_agents;
    int32_t* idx = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(work) + 256);
    float* state = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(work) + 256 + a16(total * sizeof(int32_t)));
    // no memset in front of the chain: the handle's ticket words carry the list from call to call (k_ga3c_select)
    uint32_t* ctr = reinterpret_cast<uint32_t*>(e->ga3c_ctr);
    // a launch that fails cuts the chain: the forward kernel is the one that starts the next list, so the ticket words are
    // re-zeroed (on the same stream) before the error is returned and the next call starts from an empty list again
    auto launched = [&](const char* what) -> int {
        const hipError_t s = hipGetLastError();
        if (s == hipSuccess) return CAGYM_OK;
        (void)hipMemsetAsync(e->ga3c_ctr, 0, 4 * sizeof(int32_t), st);
        return fail(e, CAGYM_E_HIP, std::string(what) + ": " + hipGetErrorString(s));
    };
    hipLaunchKernelGGL(k_ga3c_select, dim3((unsigned)((total + 1023) / 1024)), dim3(1024), 0, st, e->D, idx, ctr);
    if (int rc = launched("k_ga3c_select")) return rc;
    // the list length stays on the device: both kernels are launched for the worst case and leave beyond it
    launch_ga3c_state(e, max_observed, idx, (long long)total, ctr, state, st);
    if (int rc = launched("k_ga3c_state")) return rc;
    hipLaunchKernelGGL(k_ga3c_forward_mfma, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, weights, state, idx, 0, e->ga3c_ctr + 2, e->D.pref,
                       ext_actions, (int32_t*)nullptr, (float*)nullptr, ctr);
    if (int rc = launched("k_ga3c_forward")) return rc;
    return CAGYM_OK;
}

#ifdef CAGYM_STAMPS
int cagym_debug_stamps(unsigned long long* out16, int reset) {
    if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    }
    return 0;
}
#endif

#ifdef CAGYM_WAVETRACE
int cagym_debug_wavetrace_select(int wg) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wt_wg), &wg, sizeof(int)); }
int cagym_debug_wavetrace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wavetrace), sizeof(unsigned long long) * CAGYM_WT_STEPS * CAGYM_WT_POINTS * 8);
}
#endif

#ifdef CAGYM_WGTRACE
int cagym_debug_wgtrace(unsigned long long* out, int n_wg) {
    if (n_wg > CAGYM_WGTRACE_MAXWG) n_wg = CAGYM_WGTRACE_MAXWG;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgtrace), sizeof(unsigned long long) * CAGYM_WGTRACE_W * (size_t)n_wg);
}
#endif

int cagym_ga3c_forward(void* env, const float* weights, const float* state, const int32_t* agent_idx, int B,
                       float* ext_actions, int32_t* action_index, float* probs, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!weights || !state || !agent_idx || B < 0) return fail(e, CAGYM_E_INVALID, "bad arguments");
    if (B == 0) return CAGYM_OK;
    DEVGUARD(e);
    // 32 agents per workgroup reuse every weight 32 times; small batches take 16 so that each CU still gets >= 2 workgroups
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int which = ga3c_kernel_choice();
    if (which == GA_KERNEL_BAD) return fail(e, CAGYM_E_INVALID, "CAGYM_GA3C: unknown forward kernel (h16, mfma32 or valu)");
    if (which == GA_KERNEL_H16) {
        if (int rc = ga3c_pack(e, weights, st, false)) return rc;
        hipLaunchKernelGGL(k_ga3c_forward_h16, dim3((unsigned)((B + 31) / 32)), dim3(512), 0, st, e->ga3c_packed, state, agent_idx, B, e->D.pref,
                           ext_actions, action_index, probs);
    } else if (which == GA_KERNEL_MFMA32)
        hipLaunchKernelGGL(k_ga3c_forward_mfma, dim3((unsigned)((B + 31) / 32)), dim3(256), 0, st, weights, state, agent_idx, B,
                           (const int32_t*)nullptr, e->D.pref, ext_actions, action_index, probs, (uint32_t*)nullptr);
    else if (B <= 16 * 1024)
        hipLaunchKernelGGL(k_ga3c_forward<16>, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, st, weights, state, agent_idx, B,
                           e->D.pref, ext_actions, action_index, probs);
    else
        hipLaunchKernelGGL(k_ga3c_forward<32>, dim3((unsigned)((B + 31) / 32)), dim3(256), 0, st, weights, state, agent_idx, B,
                           e->D.pref, ext_actions, action_index, probs);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

// ---- information-gain primitives -----------------------------------------------------------------
static int ig_check(Env* e, const char* what) {
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->ig_ready) return fail(e, CAGYM_E_STATE, std::string(what) + " before cagym_ig_init");
    return CAGYM_OK;
}

int cagym_ig_init(void* env, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    if (!e) return fail(nullptr, CAGYM_E_INVALID, "null env");
    if (!e->scenarios_set) return fail(e, CAGYM_E_STATE, "cagym_ig_init before cagym_set_scenarios");
    if (e->cfg.max_obstacles <= 0 || !e->D.map_bits)
        return fail(e, CAGYM_E_UNSUPPORTED, "information-gain primitives need obstacle rasters (max_obstacles > 0)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DEVGUARD(e);
    const size_t S = e->cfg.n_scenarios, N = e->cfg.n_worlds;
    if (!e->G.d2) {
        int rc;
        if ((rc = dalloc(e, &e->G.d2, S * CAGYM_MAPD * CAGYM_MAPD)) != CAGYM_OK) return rc;
        if ((rc = dalloc(e, &e->G.belief, N * IG_BEL * IG_BEL)) != CAGYM_OK) return rc;
        if ((rc = dalloc(e, &e->G.mi, N * IG_BEL * IG_BEL)) != CAGYM_OK) return rc;
        if ((rc = dalloc(e, &e->ig_any, S)) != CAGYM_OK) return rc;
    }
    e->G.N = (int)N; e->G.S = (int)S; e->G.map_bits = e->D.map_bits; e->G.sc_nobst = e->D.sc_nobst; e->G.episode = e->D.episode;
    HIPCHK(e, hipMemsetAsync(e->ig_any, 0, S * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_ig_edt_cols, dim3((unsigned)S), dim3(320), 0, st, e->G, e->ig_any);
    hipLaunchKernelGGL(k_ig_edt_rows, dim3((unsigned)(S * CAGYM_MAPD)), dim3(320), 0, st, e->G, e->ig_any);
    hipLaunchKernelGGL(k_ig_fill_belief, dim3((unsigned)N), dim3(256), 0, st, e->G, (const uint8_t*)nullptr);
    HIPCHK(e, hipGetLastError());
    e->ig_ready = true;
    return CAGYM_OK;
}

int cagym_ig_reset_belief(void* env, const uint8_t* world_mask, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_reset_belief");
    if (rc) return rc;
    DEVGUARD(e);
    hipLaunchKernelGGL(k_ig_fill_belief, dim3((unsigned)e->cfg.n_worlds), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), e->G, world_mask);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_ig_get(void* env, uint32_t** edf_d2, double** belief) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_get");
    if (rc) return rc;
    DEVGUARD(e);
    if (edf_d2) *edf_d2 = e->G.d2;
    if (belief) *belief = e->G.belief;
    return CAGYM_OK;
}

int cagym_ig_visible_cells(void* env, const double* poses, const int32_t* world, int Q, double fov_rad, double range,
                           uint64_t* masks, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_visible_cells");
    if (rc) return rc;
    DEVGUARD(e);
    if (Q < 0 || !poses || !world || !masks) return fail(e, CAGYM_E_INVALID, "bad arguments");
    if (Q == 0) return CAGYM_OK;
    hipLaunchKernelGGL(k_ig_visible, dim3((unsigned)Q), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), e->G, poses,
                       world, fov_rad, range, reinterpret_cast<unsigned long long*>(masks));
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_ig_update_belief(void* env, const double* poses, const int32_t* n_poses, const double* detections,
                           const int32_t* n_det, int P, int Dmax, double fov_rad, double range, uint64_t* observed,
                           void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_update_belief");
    if (rc) return rc;
    DEVGUARD(e);
    if (P < 1 || Dmax < 1 || !poses || !detections || !n_det) return fail(e, CAGYM_E_INVALID, "bad arguments");
    hipLaunchKernelGGL(k_ig_update, dim3((unsigned)e->cfg.n_worlds), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       e->G, poses, n_poses, detections, n_det, P, Dmax, fov_rad, range,
                       reinterpret_cast<unsigned long long*>(observed));
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_ig_mi_reward(void* env, const uint64_t* masks, const int32_t* world, int Q, double* reward, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_mi_reward");
    if (rc) return rc;
    DEVGUARD(e);
    if (Q < 0 || !masks || !world || !reward) return fail(e, CAGYM_E_INVALID, "bad arguments");
    if (Q == 0) return CAGYM_OK;
    hipLaunchKernelGGL(k_ig_reward, dim3((unsigned)Q), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->G,
                       reinterpret_cast<const unsigned long long*>(masks), world, reward);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_ig_next_pose(void* env, const double* poses, const double* actions, const int32_t* world,
                       const double* radius, int Q, int xdt, double dt, double* next, uint8_t* feasible, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_next_pose");
    if (rc) return rc;
    DEVGUARD(e);
    if (Q < 0 || xdt < 1 || xdt > 1000 || !poses || !actions || !world || !radius || !next || !feasible)
        return fail(e, CAGYM_E_INVALID, "bad arguments");
    if (Q == 0) return CAGYM_OK;
    hipLaunchKernelGGL(k_ig_next_pose, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), e->G, poses, actions, world, radius, Q, xdt, dt, next,
                       feasible);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

int cagym_ig_rollouts(void* env, const double* pose0, const uint64_t* observed0, const uint64_t* exclude,
                      const int32_t* world, const int32_t* n_steps, const double* radius, int Q, int nsims,
                      int max_steps, int xdt, double dt, double fov_rad, double range, uint64_t seed, double* rewards,
                      uint8_t* actions, double* final_pose, uint64_t* observed_out, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_ig_rollouts");
    if (rc) return rc;
    DEVGUARD(e);
    if (Q < 0 || nsims < 1 || max_steps < 0 || max_steps > 255 || xdt < 1 || xdt > 1000 || !pose0 || !observed0 ||
        !exclude || !world || !n_steps || !radius || !rewards)
        return fail(e, CAGYM_E_INVALID, "bad arguments");
    if (Q == 0) return CAGYM_OK;
    hipLaunchKernelGGL(k_ig_rollouts, dim3((unsigned)((size_t)Q * nsims)), dim3(128), 0,
                       reinterpret_cast<hipStream_t>(stream), e->G, pose0,
                       reinterpret_cast<const unsigned long long*>(observed0),
                       reinterpret_cast<const unsigned long long*>(exclude), world, n_steps, radius, nsims, max_steps, xdt,
                       dt, fov_rad, range, (unsigned long long)seed, rewards, actions, final_pose,
                       reinterpret_cast<unsigned long long*>(observed_out));
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

namespace {
inline int dm_node_cap(const cagym_dmcts_params& p) { return 1 + 9 * (p.Ntree * p.Ncycles + 1); }
inline int dm_mask_cap(const cagym_dmcts_params& p) { return 1 + p.Ntree * p.Ncycles; }  // the root + one newly selected node per grow
inline size_t dm_align(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

size_t cagym_dmcts_workspace_bytes(int n_worlds, const cagym_dmcts_params* p) {
    if (!p || n_worlds < 1 || p->n_robots < 1 || p->Ntree < 1 || p->Ncycles < 1) return 0;
    const size_t trees = (size_t)n_worlds * p->n_robots;
    return dm_align(trees * sizeof(DmPublished)) + dm_align(trees * 2 * sizeof(int32_t)) + dm_align(trees * (size_t)dm_node_cap(*p) * sizeof(DmNode)) +
           dm_align(trees * (size_t)dm_mask_cap(*p) * sizeof(DmMasks)) + trees * (size_t)dm_node_cap(*p) * sizeof(double);  // (last: the trees' compact value arrays)
}

int cagym_dmcts_plan(void* env, const cagym_dmcts_params* params, const double* poses, void* workspace,
                     size_t workspace_bytes, double* actions, uint8_t* paths, double* stats, void* stream) {
    Env* e = reinterpret_cast<Env*>(env);
    int rc = ig_check(e, "cagym_dmcts_plan");
    if (rc) return rc;
    DEVGUARD(e);
    if (!params || !poses || !workspace || !actions || !paths || !stats) return fail(e, CAGYM_E_INVALID, "null argument");
    const cagym_dmcts_params& p = *params;
    if (p.n_robots < 1 || p.n_robots > DM_MAXR || p.horizon < 1 || p.horizon > DM_MAXH || p.Nsims < 1 || p.Nsims > DM_MAXSIMS ||
        p.comm_n < 1 || p.comm_n > DM_MAXCOMM || p.Ntree < 1 || p.Ncycles < 1 || p.xdt < 1 || p.xdt > 1000 ||
        (size_t)p.Ntree * p.Ncycles > 100000)
        return fail(e, CAGYM_E_INVALID, "Dec-MCTS parameters out of range (n_robots<=8, horizon<=8, Nsims<=32, comm_n<=8)");
    const int N = e->cfg.n_worlds;
    if (workspace_bytes < cagym_dmcts_workspace_bytes(N, params)) return fail(e, CAGYM_E_INVALID, "workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t trees = (size_t)N * p.n_robots;
    unsigned char* base = reinterpret_cast<unsigned char*>(workspace);
    DmPublished* pub = reinterpret_cast<DmPublished*>(base);
    int32_t* nn = reinterpret_cast<int32_t*>(base + dm_align(trees * sizeof(DmPublished)));
    DmNode* nodes = reinterpret_cast<DmNode*>(base + dm_align(trees * sizeof(DmPublished)) + dm_align(trees * 2 * sizeof(int32_t)));
    DmMasks* masks = reinterpret_cast<DmMasks*>(reinterpret_cast<unsigned char*>(nodes) + dm_align(trees * (size_t)dm_node_cap(p) * sizeof(DmNode)));
    double* mu = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(masks) + dm_align(trees * (size_t)dm_mask_cap(p) * sizeof(DmMasks)));
    if (p.reset_comms) HIPCHK(e, hipMemsetAsync(pub, 0, trees * sizeof(DmPublished), st));
    DmParams P;
    P.R = p.n_robots; P.Ntree = p.Ntree; P.Nsims = p.Nsims; P.horizon = p.horizon; P.Ncycles = p.Ncycles; P.comm_n = p.comm_n;
    P.node_cap = dm_node_cap(p); P.mask_cap = dm_mask_cap(p); P.xdt = p.xdt; P.call_base = p.call_base;
    P.c_p = p.c_p; P.gamma = p.gamma; P.radius = p.radius; P.dt = p.dt; P.fov = p.fov_rad; P.range = p.range; P.seed = p.seed;
    hipLaunchKernelGGL(k_dmcts_plan, dim3((unsigned)N), dim3(DM_THREADS), 0, st, e->G, P, poses, nodes, masks, mu, nn, pub, actions, paths, stats);
    HIPCHK(e, hipGetLastError());
    return CAGYM_OK;
}

}  // extern "C"
