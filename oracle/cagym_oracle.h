/* cagym_oracle.h -- CPU restatement of the reference env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C, scalar, fp64
 * restatement of gym_collision_avoidance's CollisionAvoidanceEnv.step()/reset()
 * (reference file:line cited at every function in cagym_oracle.c).  It is pinned
 * against golden vectors produced by executing the reference itself
 * (tests/golden/make_golden.py -> tests/golden/ npz files).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (libcagym_hip.so) never links, loads or falls back to it.
 *
 * Parity status: core path (a0-a11, a16) PINNED by reference-generated vectors;
 * ORCA (a12) and GA3C network (a13) "parity unpinned" (third-party rvo2 /
 * TensorFlow absent; restated from the published algorithm).
 */
#ifndef CAGYM_ORACLE_H
#define CAGYM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* policy / dynamics ids (same numbering as include/cagym.h) */
enum { CAO_POL_STATIC = 0, CAO_POL_NONCOOP = 1, CAO_POL_EXTERNAL = 2, CAO_POL_LEARNING = 3,
       CAO_POL_CARRL = 4, CAO_POL_RVO = 5, CAO_POL_GA3C = 6, CAO_POL_IGMCTS = 7 };
enum { CAO_DYN_UNICYCLE = 0, CAO_DYN_MAXTURNRATE = 1, CAO_DYN_MAXACC = 2, CAO_DYN_SECONDORDER = 3,
       CAO_DYN_FIRSTORDER = 4 };

/* game_over rule (collision_avoidance_env.py:722-736) */
enum { CAO_GO_AGENT0 = 0,      /* EVALUATE&&!HOMOGENEOUS, or TRAIN_SINGLE_AGENT: done[0]       */
       CAO_GO_ALL = 1,         /* EVALUATE&&HOMOGENEOUS: all(done)                            */
       CAO_GO_LEARNING = 2 };  /* training, multi-agent: all learning agents done (all([])=1) */

typedef struct {
    int32_t n_worlds;          /* N */
    int32_t max_agents;        /* M: agent slots per world (Config.MAX_NUM_AGENTS_IN_ENVIRONMENT) */
    int32_t max_obstacles;     /* K rectangles per world (0 = free space) */
    int32_t game_over_mode;    /* CAO_GO_* */
    int32_t collide_with_static; /* Config.COLLISION_AV_W_STATIC_AGENT (config.py:52) */
    int32_t laserscan;         /* 1: agents carry LaserScanSensor */
    double dt;                 /* Config.DT (config.py:29) */
    int32_t rvo_max_neighbors; /* RVO maxNeighbors; 0 = max_agents (Config.MAX_NUM_AGENTS_IN_ENVIRONMENT, RVOPolicy.py:15) */
    int32_t reserved;
} cao_config;
#define CAO_MAXOBST 64 /* rectangles per world the ORCA obstacle code looks at */

typedef struct cao_env cao_env;

cao_env* cao_create(const cao_config* cfg);
void cao_destroy(cao_env* e);

/* agents6[N,M,6] = sx,sy,gx,gy,pref_speed,radius; heading0[N,M] or NULL (toward goal,
 * agent.py:29-31); policy_id/dynamics_id[N,M]; n_agents[N] or NULL (=M);
 * coop[N,M] or NULL (1.0, agent.py:10); obstacles[N,K,4] = xl,yl,xu,yu; n_obst[N]. */
void cao_set_scenario(cao_env* e, const double* agents6, const double* heading0, const int32_t* policy_id,
                      const int32_t* dynamics_id, const int32_t* n_agents, const double* coop,
                      const double* obstacles, const int32_t* n_obst);
/* world_mask[N] (NULL = all): re-initialise the masked worlds from the scenario and sense. */
void cao_reset(cao_env* e, const uint8_t* world_mask);
/* ext_actions[N,M,2] doubles (may be NULL when no agent needs one). */
void cao_step(cao_env* e, const double* ext_actions);
void cao_run(cao_env* e, int n_steps); /* n_steps x (step all worlds, restart finished ones) */
int cao_set_threads(int n);            /* OpenMP team size of the world loops (n <= 0: leave); returns the size in force */

/* zero-copy views for the tests; field ids below */
enum { CAO_F_POS = 0, CAO_F_VEL, CAO_F_HEADING, CAO_F_SPEED, CAO_F_DELTA_HEADING, CAO_F_DIST_TO_GOAL,
       CAO_F_PAST_DIST_TO_GOAL, CAO_F_HEADING_EGO, CAO_F_VEL_EGO, CAO_F_REF_PRLL, CAO_F_REL_GOAL,
       CAO_F_TIME_REMAINING, CAO_F_T, CAO_F_PAST_ACTIONS, CAO_F_REWARD, CAO_F_OAS, CAO_F_LASERSCAN,
       CAO_F_ACTION, CAO_F_COUNT };
enum { CAO_U_IS_AT_GOAL = 0, CAO_U_WAS_AT_GOAL, CAO_U_IN_COLLISION, CAO_U_WAS_IN_COLLISION,
       CAO_U_RAN_OUT_OF_TIME, CAO_U_IS_DONE, CAO_U_GAME_OVER, CAO_U_MAP, CAO_U_COUNT };
enum { CAO_I_STEP_NUM = 0, CAO_I_NUM_OBSERVED, CAO_I_COUNT };
double* cao_f64(cao_env* e, int field);
uint8_t* cao_u8(cao_env* e, int field);
int32_t* cao_i32(cao_env* e, int field);

/* ---- stand-alone primitives (also used by the env above) ---- */
/* Map.get_occupancy_grid (Map.py:107-123): 300x300 u8 raster from rectangles. */
void cao_rasterize(const double* obstacles, int n_obst, uint8_t* map300);
/* ORCA ego solve (RVOPolicy.py:53-117 + RVO2 Agent::computeNewVelocity), fp32 inside. */
void cao_orca_action(int M, int ego, const double* pos, const double* vel, const double* goal,
                     const double* pref_speed, const double* radius, double heading, double collab,
                     double dt, double* action_out);
/* the same with RVO maxNeighbors and the world's rectangles rects[n_obst][4] = xl, yl, xu, yu (RVOPolicy.py:56-57;
 * obstacle half of Agent::computeNewVelocity, numObstLines protected in linearProgram3).  Optional outputs:
 * new_vel_out[2] fp32 velocity chosen by the linear programs, lines_out[n][4] half-planes (point, direction) in solve
 * order, n_lines_out[2] = {numObstLines, total lines}. */
void cao_orca_action_ex(int M, int ego, const double* pos, const double* vel, const double* goal,
                        const double* pref_speed, const double* radius, double heading, double collab,
                        double dt, int max_neighbors, const double* rects, int n_obst, double* action_out,
                        float* new_vel_out, float* lines_out, int* n_lines_out);

/* The two halves of the above, separately (tests/golden/rvo2_standin.py runs the unmodified reference RVOPolicy on top
 * of cao_rvo2_step_agent; tests compare cao_rvo_sim_inputs with the setter arguments the reference recorded):
 * what RVOPolicy.py:63-85 hands to the simulator (float-narrowed), and what doStep() makes of it for one agent. */
void cao_rvo_sim_inputs(int M, int ego, const double* pos, const double* vel, const double* goal,
                        const double* pref_speed, const double* radius, double collab,
                        float* p32, float* v32, float* r32, float* pref_vel2, float* max_speed, float* collab32);
void cao_rvo2_step_agent(int n, int ego, const float* pos, const float* vel, const float* radius,
                         const float* pref_vel2, float max_speed, float collab, float neighbor_dist, int max_neighbors,
                         float time_horizon, float time_horizon_obst, float time_step, const double* rects, int n_obst,
                         float* new_pos2, float* new_vel2, float* lines_out, int* n_lines_out);

/* GA3C-CADRL state vectors out[N,M,76] (policies/GA3CCADRLPolicy.py:45-106) */
void cao_ga3c_states(cao_env* e, int max_observed, double* out);

/* ---- information-gain primitives (cagym_oracle_ig.c); one world at a time ---- */
/* edfMap.update (information_models/edfMap.py:11-12): edf[300*300] f64 and/or d2[300*300] u32 */
void cao_edt(const uint8_t* map300, double* edf, uint32_t* d2);
/* edfMap.checkVisibility (edfMap.py:21-47) */
int cao_ig_check_visibility(const double* edf, const double* pose2, const double* goal2);
/* targetMap.getVisibleCells (information_models/targetMap.py:43-84): mask[60] u64, bit i of mask[j] */
int cao_ig_visible(const double* edf, const double* pose3, double fov_rad, double range, uint64_t* mask);
/* targetMap.update (targetMap.py:86-128) */
void cao_ig_update(double* belief, const double* edf, int P, const double* poses, const int32_t* ndet,
                   const double* dets, int Dmax, double fov_rad, double range, uint64_t* observed);
/* targetMap.get_reward_from_cells (targetMap.py:130-143) */
double cao_ig_reward(const double* belief, const uint64_t* mask);
/* ig_mcts.get_next_pose (policies/ig_mcts.py:154-183) */
int cao_ig_next_pose(const double* edf, const double* pose3, const double* action2, int xdt, double dt, double radius,
                     double* next3);
uint32_t cao_ig_rand_primitive(uint64_t seed, uint32_t q, uint32_t sim, uint32_t step);
/* Tree._simulate random roll-out (pydecmcts/DecMCTS.py:233-271) + mcts_reward (ig_mcts.py:234-241) */
double cao_ig_rollout(const double* belief, const double* edf, const double* pose0, const uint64_t* observed0,
                      const uint64_t* exclude, int n_steps, int xdt, double dt, double radius, double fov_rad,
                      double range, uint64_t seed, uint32_t q, uint32_t sim, uint8_t* actions_out, double* pose_out,
                      uint64_t* observed_out);

#ifdef __cplusplus
}
#endif
#endif
