/* cagym.h -- C ABI of the MI355X-native batched CollisionAvoidanceEnv (libcagym_hip.so).
 *
 * The reference (mlodel/gym-exploration-2d) has no FFI layer: its boundary is the Python
 * gym.Env object `CollisionAvoidanceEnv` (gym_collision_avoidance/envs/collision_avoidance_env.py,
 * "env.py" below).  Each entry point here cites the reference interface it replaces; the Python
 * host (gym-exploration-2d_amd/) binds them with ctypes and mirrors the gym.Env surface.
 *
 * Conventions
 *  - Plain pointers and sizes only.  Pointers marked DEVICE are HIP device pointers (e.g.
 *    torch.Tensor.data_ptr() of a contiguous ROCm tensor); HOST pointers are ordinary memory.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *    all work is enqueued on it, nothing inside cagym_step / cagym_rollout synchronises the host.
 *  - Every call returns 0 or a negative CAGYM_E* code; cagym_last_error() gives the message.
 *    Nothing throws across the ABI.  There is NO CPU fallback: without a HIP device
 *    cagym_create fails with CAGYM_E_NODEVICE.
 *  - A handle is not thread-safe; one handle per GPU; at most ONE call per handle in flight on the device unless a function
 *    says otherwise (calls on different streams must be ordered by the caller: several entry points keep cursors on the handle).
 *  - Layout: N worlds x M agent slots, structure-of-arrays, world-major ([N][M]).
 */
#ifndef CAGYM_H
#define CAGYM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAGYM_VERSION 112 /* 0.1.2: cagym_step_begin / cagym_step_finish, CAGYM_E_DEVICE */

enum { CAGYM_OK = 0, CAGYM_E_INVALID = -1, CAGYM_E_NODEVICE = -2, CAGYM_E_HIP = -3, CAGYM_E_NOMEM = -4,
       CAGYM_E_STATE = -5, CAGYM_E_UNSUPPORTED = -6,
       CAGYM_E_DEVICE = -7 /* a kernel of an earlier launch on this handle reported an internal error (a bounded intra-workgroup wait
                              expired): the handle's device state is void; every later launching call returns this code */ };

/* policy ids: which action map / policy drives an agent (env.py:298-320) */
enum { CAGYM_POL_STATIC = 0,   /* policies/StaticPolicy.py:9-12            a = (0, 0)                      */
       CAGYM_POL_NONCOOP = 1,  /* policies/NonCooperativePolicy.py:10-13   a = (pref_speed, -heading_ego)  */
       CAGYM_POL_EXTERNAL = 2, /* raw (speed, delta_heading) from ext_actions                  */
       CAGYM_POL_LEARNING = 3, /* policies/LearningPolicy.py:11-16         a = (v_pref*u0, 4*(2*u1-1))     */
       CAGYM_POL_CARRL = 4,    /* policies/CARRLPolicy.py:5-15             11-row table, index in ext[.,0] */
       CAGYM_POL_RVO = 5,      /* policies/RVOPolicy.py:53-117             ORCA half-planes (other agents and, :56-57, the
                                  world's rectangles) + 2-D LP; needs 2*max_obstacles + max_agents - 1 <= 32 (64 for
                                  max_agents > 10) half-planes per ego and non-degenerate rectangles               */
       CAGYM_POL_GA3C = 6,     /* policies/GA3CCADRLPolicy.py:34-43        action supplied by cagym_ga3c_* */
       CAGYM_POL_IGMCTS = 7 }; /* policies/ig_mcts.py:79-109               (v, omega) supplied by planner  */

/* dynamics ids (envs/dynamics/) */
enum { CAGYM_DYN_UNICYCLE = 0,    /* UnicycleDynamics.py:10-24                 */
       CAGYM_DYN_MAXTURNRATE = 1, /* UnicycleDynamicsMaxTurnRate.py:11-25      */
       CAGYM_DYN_MAXACC = 2,      /* UnicycleDynamicsMaxAcc.py:17-39           */
       CAGYM_DYN_SECONDORDER = 3, /* UnicycleSecondOrderEulerDynamics.py:12-29 */
       CAGYM_DYN_FIRSTORDER = 4 };/* FirstOrderDynamics.py:10-23               */

/* game_over rule (env.py:722-736) */
enum { CAGYM_GO_AGENT0 = 0,    /* EVALUATE_MODE && !HOMOGENEOUS_TESTING, or TRAIN_SINGLE_AGENT */
       CAGYM_GO_ALL = 1,       /* EVALUATE_MODE && HOMOGENEOUS_TESTING                         */
       CAGYM_GO_LEARNING = 2 };/* all agents with CAGYM_POL_LEARNING done                      */

/* per-agent status byte (`flags` output and state view) */
enum { CAGYM_FLAG_AT_GOAL = 1, CAGYM_FLAG_IN_COLLISION = 2, CAGYM_FLAG_RAN_OUT_OF_TIME = 4, CAGYM_FLAG_DONE = 8,
       CAGYM_FLAG_WAS_AT_GOAL = 16, CAGYM_FLAG_WAS_IN_COLLISION = 32, CAGYM_FLAG_ACTIVE = 64 };

/* Replaces the class attributes of envs/config.py read by the hot path. */
typedef struct {
    int32_t n_worlds;            /* N: worlds stepped together (one reference env instance each)            */
    int32_t max_agents;          /* M: Config.MAX_NUM_AGENTS_IN_ENVIRONMENT (config.py:70); 2..32           */
    int32_t n_scenarios;         /* S >= N: scenario pool; world w starts episode e on scenario (w+e*N)%S   */
    int32_t max_obstacles;       /* rectangles per scenario (0 = free space; env.py:492-497)                */
    int32_t game_over_mode;      /* CAGYM_GO_*                                                              */
    int32_t collide_with_static; /* Config.COLLISION_AV_W_STATIC_AGENT (config.py:52)                       */
    int32_t laserscan;           /* 1: every agent carries a LaserScanSensor (sensors/LaserScanSensor.py)   */
    int32_t device;              /* HIP device ordinal                                                      */
    double dt;                   /* Config.DT (config.py:29)                                                */
    int32_t rvo_max_neighbors;   /* RVO maxNeighbors; 0 = max_agents, as policies/RVOPolicy.py:15,25 passes it */
    int32_t reserved;            /* 0 */
} cagym_config;

#define CAGYM_EGO_WIDTH 12
/* Caller-owned DEVICE output buffers of one step.  Any pointer may be NULL (not written).
 * obs_ego columns: dist_to_goal, rel_goal.x, rel_goal.y, radius, heading_ego_frame,
 *   heading_global_frame, pos.x, pos.y, pref_speed, num_other_agents_observed, use_ppo, 0
 *   (the scalar keys of Config.STATE_INFO_DICT, config.py:104-215).
 * obs_oas: OtherAgentsStatesSensor table, rows farthest->closest (sensors/OtherAgentsStatesSensor.py:11-77).
 * For cagym_rollout every buffer has a leading [T] dimension. */
typedef struct {
    float* obs_oas;     /* [N, M, M-1, 10] f32 */
    float* obs_ego;     /* [N, M, 12]      f32 */
    float* laserscan;   /* [N, M, 16]      f32 (only when cfg.laserscan)                           */
    float* reward;      /* [N, M]          f32: _compute_rewards (env.py:502-567), all agents      */
    uint8_t* flags;     /* [N, M]          u8 : CAGYM_FLAG_* after the step (env.py:711-721)       */
    uint8_t* game_over; /* [N]             u8 : env.py:722-738                                     */
} cagym_outputs;

/* Zero-copy DEVICE views of the SoA state (for parity tests / Agent-like host views; agent.py:9-109). */
typedef struct {
    double *pos_x, *pos_y, *vel_x, *vel_y, *heading, *heading_ego, *dist_to_goal, *time_remaining, *t;
    double *goal_x, *goal_y, *radius, *pref_speed, *speed, *delta_heading, *aux0, *aux1;
    float* action;          /* [N, M, 2] last applied (speed, delta_heading), fp32 as env.py:289 */
    uint32_t* status;       /* [N, M] bits 0-7 CAGYM_FLAG_*, 8-11 policy id, 12-15 dynamics id   */
    int32_t* step_num;      /* [N, M] agent.py:186                                               */
    int32_t* n_agents;      /* [N]                                                               */
    int32_t* n_observed;    /* [N, M] num_other_agents_observed                                  */
    int32_t* episode;       /* [N] episodes started by this world (selects the scenario)         */
    uint32_t* map_bits;     /* [S, 300, 10] bit-packed occupancy rasters (Map.py:107-123) or NULL */
    /* cumulative per-world episode statistics (the payload of the multi-GPU all-gather) */
    float* stat_return;     /* [N] sum over finished episodes of agent-0 episode return          */
    int32_t* stat_episodes; /* [N] finished episodes                                             */
    int32_t* stat_steps;    /* [N] env steps in finished episodes                                */
    int32_t* stat_outcomes; /* [N, 3] agents finished at goal / in collision / timed out         */
} cagym_state_ptrs;

int cagym_version(void);

/* CollisionAvoidanceEnv.__init__ (env.py:60-160) for N worlds. */
int cagym_create(const cagym_config* cfg, void** env_out);
/* CollisionAvoidanceEnv.close (env.py:268-270). */
int cagym_destroy(void* env);
const char* cagym_last_error(void* env);

/* set_agents / _init_agents / set_static_map (env.py:387-388, 403-476, 478-500): upload the scenario
 * pool.  All pointers HOST.  agents6[S,M,6] = start_x, start_y, goal_x, goal_y, pref_speed, radius (the
 * "legacy cadrl" row of test_cases.py:1970-2014); heading0[S,M] or NULL (toward goal, agent.py:29-31);
 * policy_id / dynamics_id [S,M]; n_agents[S] or NULL (= M); coop[S,M] or NULL (1.0, agent.py:10);
 * obstacles[S,K,4] = xl, yl, xu, yu or NULL; n_obst[S].  Rasterises the obstacle maps on device. */
int cagym_set_scenarios(void* env, const double* agents6, const double* heading0, const int32_t* policy_id,
                        const int32_t* dynamics_id, const int32_t* n_agents, const double* coop,
                        const double* obstacles, const int32_t* n_obst, void* stream);

/* reset() (env.py:234-266) for the worlds whose world_mask byte is non-zero (DEVICE [N], NULL = all):
 * re-initialise agents from the world's current scenario, sense, write observations to `out`.
 * advance_episode != 0 moves the masked worlds to their next scenario first. */
int cagym_reset(void* env, const uint8_t* world_mask, int advance_episode, const cagym_outputs* out, void* stream);

/* step(actions) (env.py:162-232): _take_action -> _compute_rewards(+_check_for_collisions) -> _get_obs ->
 * _check_which_agents_done.  ext_actions DEVICE [N,M,2] f32 or NULL (all agents internal, env_utils.py:46). */
int cagym_step(void* env, const float* ext_actions, const cagym_outputs* out, void* stream);

/* cagym_step followed, in the same launch, by DummyVecEnv's auto-reset (exp/env_utils.py:29-31): a world whose
 * game_over fires restarts on its next scenario, its episode statistics are folded, and the observation written
 * for this step -- OtherAgentsStates, scalar keys and, with cfg.laserscan, the laser scan (_get_obs runs every sensor,
 * env.py:740-753) -- is the first one of the new episode (reward / flags / game_over are the terminal ones). */
int cagym_step_autoreset(void* env, const float* ext_actions, const cagym_outputs* out, void* stream);

/* step(actions) in TWO launches, for callers whose external actions come from a device policy of their own (cfg4: cagym_ga3c_act).
 * _take_action (env.py:287-340) gathers the actions of ALL agents before any agent moves, and an internal RVO policy
 * (policies/RVOPolicy.py:53-117) reads the state BEFORE the step only - so the RVO half of the step does not have to wait for the
 * external actions:
 *   cagym_step_begin   the ORCA half-planes (agents and rectangles) and linear programs of every live RVO ego on the current state;
 *                      8 bytes per agent are kept on the handle.  Enqueue it on a stream of its own, beside the policy that
 *                      produces ext_actions; it reads the state, never writes it.  A no-op for handles without RVO agents.
 *   cagym_step_finish  the rest of the step (action maps + dynamics with every action in hand, collisions, rewards, done /
 *                      game_over, observations, and - auto_reset != 0 - cagym_step_autoreset's restart).  The caller orders it
 *                      behind BOTH the begin launch and the producer of ext_actions (stream / event dependencies).
 * begin + finish produce bit for bit what cagym_step / cagym_step_autoreset produce (tests/test_split_step.py).  finish without
 * a begin since the last finish / step / reset / rollout returns CAGYM_E_STATE.  Generation-3 kernels only. */
int cagym_step_begin(void* env, void* stream);
int cagym_step_finish(void* env, const float* ext_actions, const cagym_outputs* out, int auto_reset, void* stream);

/* n_steps consecutive step() calls in ONE launch for worlds whose agents are all driven internally
 * (Static / NonCooperative / RVO): the agent records stay on chip, every step writes its outputs (laserscan
 * [T, N, M, 16] included when cfg.laserscan) to slice t of `out` ([T, ...] buffers; T = n_steps).  auto_reset != 0: a world whose game_over fires is
 * reset onto its next scenario inside the kernel (what stable-baselines' DummyVecEnv does around the
 * reference env, exp/env_utils.py:29-31) and its episode statistics are accumulated. */
int cagym_rollout(void* env, int n_steps, int auto_reset, const cagym_outputs* out, void* stream);

int cagym_get_state(void* env, cagym_state_ptrs* out);

/* The per-world episode statistics as ONE packed [N, 6] int32 table, written by one kernel: {bit pattern of the fp32 return
 * sum, finished episodes, env steps in them, agents at goal / in collision / timed out} - the 24-byte record the multi-GPU
 * all-gather carries (the reference's per-episode statistics are experiments/src/env_utils.py:41-75).
 * records: device pointer, N * 6 int32. */
int cagym_pack_episode_stats(void* env, int32_t* records, void* stream);

/* Name of the kernel instantiation this handle launches for cagym_step / cagym_step_autoreset (rollout == 0) or
 * cagym_rollout (rollout != 0), as rocprofv3 --kernel-trace prints it (bench.py reports it next to the roofline). */
int cagym_kernel_name(void* env, int rollout, int auto_reset, char* buf, int buf_len);

/* LaserScanSensor.sense (sensors/LaserScanSensor.py:27-58) on the current state -> laserscan [N,M,16]. */
int cagym_laserscan(void* env, float* laserscan, void* stream);

/* GA3CCADRLPolicy.agents_to_ga3c_cadrl_state (policies/GA3CCADRLPolicy.py:45-106) for every active agent:
 * state DEVICE [N,M,76] f32 = [id, n_others, dist_to_goal, heading_ego, pref_speed, radius, 10 x (p_prll,
 * p_orth, v_prll, v_orth, r_other, r_host+r_other, edge distance)], others sorted by (-round(d,2), p_orth),
 * the closest `max_observed` (Config.MAX_NUM_OTHER_AGENTS_OBSERVED, <= 10) kept, farthest first.  The network
 * itself (LSTM-64 + 3 x FC-256, network.py:65-98) is evaluated by the host policy on state[:, :, 1:]. */
int cagym_ga3c_state(void* env, int max_observed, float* state, void* stream);

/* NetworkVP_rnn forward pass (policies/GA3C_CADRL/network.py:65-98) + argmax + action table (network.py:8-17,
 * GA3CCADRLPolicy.find_next_action :34-43) for the B agents listed in agent_idx (flat world*M + slot), reading
 * their rows of `state` (cagym_ga3c_state).  weights: DEVICE blob of CAGYM_GA3C_NWEIGHTS floats in the order
 * lstm kernel [71,256], lstm bias, layer1 kernel [68,256], bias, layer2 kernel [256,256], bias, fullyconnected1
 * kernel [256,256], bias, logits_p kernel [256,11], bias (TensorFlow [in][out] layout).  Writes
 * ext_actions[agent] = (pref_speed * a0, a1) (DEVICE [N,M,2] f32); action_index [B] i32 and probs [B,11] f32
 * (softmax_p) are optional. */
#define CAGYM_GA3C_NWEIGHTS 170507
/* The forward kernels multiply on gfx950's 16-bit matrix cores with every fp32 operand split into two f16 halves (three matrix
 * instructions per product, fp32 accumulation: fp32-class accuracy, csrc/cagym_ga3c16.h).  The handle keeps the blob re-ordered
 * into operand fragments; that copy is made on `stream` the first time cagym_ga3c_forward / cagym_ga3c_act see a blob ADDRESS.
 * A caller that rewrites the same blob in place (training) calls cagym_ga3c_load_weights afterwards; one blob per handle is
 * cached.  CAGYM_GA3C=mfma32 / valu (environment, read per call) select the exact-fp32 kernels of rounds 2 / 1 for A/B. */
int cagym_ga3c_load_weights(void* env, const float* weights, void* stream);
int cagym_ga3c_forward(void* env, const float* weights, const float* state, const int32_t* agent_idx, int B,
                       float* ext_actions, int32_t* action_index, float* probs, void* stream);

/* GA3CCADRLPolicy.find_next_action (policies/GA3CCADRLPolicy.py:34-43) for EVERY active agent whose policy id is
 * CAGYM_POL_GA3C, in one call and without a host round trip: the agents are listed on the device, their state vectors built
 * (only theirs, as the reference does per agent), the network evaluated and (pref_speed * a0, a1) written to their rows of
 * ext_actions [N*M, 2] f32; other rows are untouched.  ONE kernel launch (round 4: a workgroup per 32 worlds lists its agents,
 * keeps their state vectors in LDS and runs the network on them); it replays from a captured HIP graph.  work: caller-owned
 * DEVICE scratch of cagym_ga3c_act_workspace_bytes(env) bytes - used (overwritten; layout private) only by the A/B kernels
 * CAGYM_GA3C=mfma32 / valu, which run the three-launch chain of rounds 2 - 3; one call per handle in flight. */
size_t cagym_ga3c_act_workspace_bytes(void* env);
int cagym_ga3c_act(void* env, const float* weights, int max_observed, void* work, float* ext_actions, void* stream);

/* ---- information-gain planner primitives (cfg 5).  All pointers DEVICE.  A visibility set is a
 * [60] u64 mask: bit i of word j <=> belief cell (i, j) (x index i, y index j; 0.5 m cells over 30x30 m). ---- */

/* ig_mcts.set_param (policies/ig_mcts.py:56-77): build the Euclidean distance field of every scenario
 * raster (edfMap.update, information_models/edfMap.py:11-12) and N belief grids at prior 1.0
 * (targetMap.__init__, information_models/targetMap.py:7-24).  Needs max_obstacles > 0. */
int cagym_ig_init(void* env, void* stream);
/* re-initialise the belief grids of the masked worlds (NULL = all) to the prior. */
int cagym_ig_reset_belief(void* env, const uint8_t* world_mask, void* stream);
/* READ-ONLY views: edf_d2 [S,300,300] u32 squared cell distances (EDF = sqrt(d2)*0.1), belief [N,60,60] f64 odds.  The belief
 * changes through cagym_ig_reset_belief / cagym_ig_update_belief only: the library keeps the per-cell mutual information of the
 * belief beside it (the reward sums of cagym_ig_mi_reward / cagym_ig_rollouts / cagym_dmcts_plan read that cache). */
int cagym_ig_get(void* env, uint32_t** edf_d2, double** belief);
/* targetMap.getVisibleCells (targetMap.py:43-84) for Q poses (x, y, phi) of worlds world[q]. */
int cagym_ig_visible_cells(void* env, const double* poses, const int32_t* world, int Q, double fov_rad,
                           double range, uint64_t* masks, void* stream);
/* targetMap.update (targetMap.py:86-128), frame='global': poses [N,P,3] applied in order, n_poses [N] or
 * NULL (= P), detections [N,P,Dmax,2] global positions, n_det [N,P]; observed [N,60] = union of the visible
 * sets (NULL to skip).  ig_mcts.update_belief (ig_mcts.py:117-133). */
int cagym_ig_update_belief(void* env, const double* poses, const int32_t* n_poses, const double* detections,
                           const int32_t* n_det, int P, int Dmax, double fov_rad, double range, uint64_t* observed,
                           void* stream);
/* targetMap.get_reward_from_cells (targetMap.py:130-143): reward[q] = sum of cell MI over masks[q]. */
int cagym_ig_mi_reward(void* env, const uint64_t* masks, const int32_t* world, int Q, double* reward, void* stream);
/* ig_mcts.get_next_pose (ig_mcts.py:154-183): xdt Euler sub-steps of dt; feasible[q] = 0 where the
 * reference returns None (next[q] is then the input pose). */
int cagym_ig_next_pose(void* env, const double* poses, const double* actions, const int32_t* world,
                       const double* radius, int Q, int xdt, double dt, double* next, uint8_t* feasible, void* stream);
/* Tree._simulate (policies/pydecmcts/DecMCTS.py:233-271) x nsims per query: n_steps[q] uniformly random
 * motion primitives (ig_mcts.mcts_avail_actions :247-253; counter-based RNG on (seed, q, sim, step)) from
 * pose0[q] with already-observed set observed0[q]; reward = MI(observed minus exclude[q]) (mcts_reward :234-241).
 * rewards [Q,nsims]; actions [Q,nsims,max_steps] primitive index 0..8 or 255 (infeasible draw; may be NULL);
 * final_pose [Q,nsims,3] (may be NULL); observed_out [Q,nsims,60] = cells observed along each roll-out including
 * observed0, before exclusion (may be NULL) -- the set a robot communicates to its team (MCTS_state.obsvd_cells). */
int cagym_ig_rollouts(void* env, const double* pose0, const uint64_t* observed0, const uint64_t* exclude,
                      const int32_t* world, const int32_t* n_steps, const double* radius, int Q, int nsims,
                      int max_steps, int xdt, double dt, double fov_rad, double range, uint64_t seed,
                      double* rewards, uint8_t* actions, double* final_pose, uint64_t* observed_out, void* stream);

/* OccupancyGridSensor.sense (sensors/OccupancyGridSensor.py:70-98) for every agent of the current
 * state: grid DEVICE [N, M, 60, 60] u8 (0 / 1), the 'local_grid' observation: the obstacle raster rotated into the
 * agent's heading (cv2.warpAffine restated, bilinear, zero border) and cropped around the agent.  Needs
 * max_obstacles > 0.  PARITY UNPINNED: OpenCV is not available to check the restatement against. */
int cagym_occupancy_grid(void* env, uint8_t* grid, void* stream);

/* ---- on-device scenario generation ---------------------------------------------------
 * train_agents_random_positions (test_cases.py:1362-1463) for every scenario of the pool, one lane per
 * scenario: per agent draw start x, y and goal x, y ~ U(-side, side) (four draws per attempt, in that order)
 * until the start is >= min_sep from every earlier start, the goal >= min_sep from every earlier goal
 * (is_pose_valid, test_cases.py:129-133) and |goal - start| >= min_travel; radius / pref_speed / cooperation
 * coefficient constant (:1364-1365, :1426).  Agents per scenario ~ U{n_min..n_max} (random.randint(2, n) when
 * unseeded, n when seeded, :1367-1372).  Agent 0 gets (ego_policy, ego_dynamics); every other agent policy_b
 * with probability p_b else policy_a (random.choice of two = 0.5, :1417; 0.2 in :1352-1355) and other_dynamics.
 * Random numbers: counter-based (splitmix64 finaliser on (seed, scenario, draw)), NOT numpy's global MT19937
 * stream, so agreement with the reference is distributional; the CPU oracle uses the same generator and agrees
 * bit for bit.  max_tries bounds the rejection loop (the last draw is kept; *n_failed counts such agents). */
typedef struct cagym_gen_params {
    uint64_t seed;
    int32_t n_min, n_max;
    int32_t ego_policy, ego_dynamics;
    int32_t policy_a, policy_b, other_dynamics;
    int32_t max_tries;
    double p_b;
    double side, min_travel, min_sep, radius, pref_speed, coop;
} cagym_gen_params;
int cagym_generate_scenarios(void* env, const cagym_gen_params* params, int32_t* n_failed_host, void* stream);

/* zero-copy DEVICE views of the scenario pool (parity tests, dataset export) */
typedef struct cagym_scenario_ptrs {
    const double* agents6;    /* [S, M, 6] */
    const int32_t* policy;    /* [S, M]    */
    const int32_t* dynamics;  /* [S, M]    */
    const int32_t* n_agents;  /* [S]       */
    const double* coop;       /* [S, M]    */
} cagym_scenario_ptrs;
int cagym_get_scenarios(void* env, cagym_scenario_ptrs* out);

/* ---- Dec-MCTS planning step on the device ------------------------------------------------
 * ig_mcts.find_next_action for every IG robot of every world (ig_mcts.py:79-109) with the tree of
 * pydecmcts/DecMCTS.py:92-360 kept on the device: per cycle and robot, Ntree times { sample one communicated
 * plan per robot listened to, UCT selection, expansion by the feasible motion primitives, Nsims random roll-outs
 * to the horizon, discounted back-propagation, top-comm_n action distribution (q = mu^2) }, then publish.  One
 * workgroup per world; same generator keys, summation orders and tie rules as the host planner
 * gym-exploration-2d_amd/dmcts.py, whose decisions it reproduces.  poses DEVICE [N,R,3] (x, y, heading) of the
 * IG robots; workspace DEVICE, caller-owned, cagym_dmcts_workspace_bytes() bytes, holds the trees and the plans
 * the robots communicated (kept across calls like policy.best_paths; reset_comms != 0 forgets them, e.g. at an
 * episode start); call_base = number of tree grows requested by earlier calls with this seed (keys the random
 * streams).  Outputs DEVICE: actions [N,R,2] = (v, omega) of the first step of each robot's best path,
 * paths [N,R,8] primitive indices of that path (254 = infeasible draw, 255 = none), stats [N,R,3] = root mu,
 * root N, node count (may be NULL).  Limits: n_robots <= 8, horizon <= 8, Nsims <= 32, comm_n <= 8. */
typedef struct cagym_dmcts_params {
    int32_t n_robots, Ntree, Nsims, horizon, Ncycles, comm_n, xdt, reset_comms;
    uint32_t call_base, pad;
    double c_p, gamma, radius, dt, fov_rad, range;
    uint64_t seed;
} cagym_dmcts_params;
size_t cagym_dmcts_workspace_bytes(int n_worlds, const cagym_dmcts_params* params);
int cagym_dmcts_plan(void* env, const cagym_dmcts_params* params, const double* poses, void* workspace,
                     size_t workspace_bytes, double* actions, uint8_t* paths, double* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif
