"""gym.Env-style facade (one world) over the batched HIP env.

Keeps the surface the reference's drivers use (collision_avoidance_env.py:60-232, 387-388, 776-781;
experiments/src/example.py:22-54, experiments/src/env_utils.py:34-62):
    env = CollisionAvoidanceEnv(); env.set_agents(agents); obs = env.reset()
    obs, rewards, game_over, info = env.step(actions)
with `obs[i][state]` for state in Config.STATES_IN_OBS, `info['which_agents_done'][agent.id]`,
`env.agents[i].pos_global_frame` ..., `env.prev_episode_agents`, `env.observation_space`,
`env.action_space`.  Deviations D1-D4 of SURVEY.md section 8 apply (set_agents is honoured, no
ig_mcts agent is fine, LearningPolicy gets (agent, actions[i]), reset needs no targetMap).
`gym` itself is not required: spaces are the minimal Box/Dict below.
"""
import copy

import numpy as np

from . import scenarios as sc


# ---- markers with the reference's class names (envs/policies, envs/dynamics, envs/sensors) -------------
class _Policy(object):
    policy_id = None
    str = "NoPolicy"
    is_still_learning = False
    is_external = False

    def __str__(self):
        return self.str


class StaticPolicy(_Policy):
    policy_id, str = sc.POLICY_STATIC, "Static"


class NonCooperativePolicy(_Policy):
    policy_id, str = sc.POLICY_NONCOOP, "NonCooperativePolicy"


class ExternalPolicy(_Policy):
    policy_id, str, is_external = sc.POLICY_EXTERNAL, "External", True


class LearningPolicy(_Policy):
    policy_id, str, is_still_learning = sc.POLICY_LEARNING, "learning", True


class CARRLPolicy(ExternalPolicy):
    policy_id, str = sc.POLICY_CARRL, "CARRL"


class RVOPolicy(_Policy):
    policy_id, str = sc.POLICY_RVO, "RVO"


class GA3CCADRLPolicy(_Policy):
    policy_id, str = sc.POLICY_GA3C, "GA3C_CADRL"


class ig_mcts(_Policy):
    policy_id, str = sc.POLICY_IGMCTS, "ig_mcts"


class UnicycleDynamics(object):
    dynamics_id = sc.DYN_UNICYCLE


class UnicycleDynamicsMaxTurnRate(object):
    dynamics_id = sc.DYN_MAXTURNRATE


class UnicycleDynamicsMaxAcc(object):
    dynamics_id = sc.DYN_MAXACC


class UnicycleSecondOrderEulerDynamics(object):
    dynamics_id = sc.DYN_SECONDORDER


class FirstOrderDynamics(object):
    dynamics_id = sc.DYN_FIRSTORDER


class OtherAgentsStatesSensor(object):
    name = "other_agents_states"


class LaserScanSensor(object):
    name = "laserscan"


# ---- Config: the class-attribute flags of envs/config.py that the hot path reads -------------------------
class Config(object):
    DT = 0.1
    EVALUATE_MODE = False
    PLAY_MODE = False
    TRAIN_SINGLE_AGENT = True
    HOMOGENEOUS_TESTING = False
    COLLISION_AV_W_STATIC_AGENT = False
    MAX_NUM_AGENTS_IN_ENVIRONMENT = 10
    LASERSCAN_LENGTH = 16
    NEAR_GOAL_THRESHOLD = 0.75
    MAX_TIME_RATIO = 3.0
    GETTING_CLOSE_RANGE = 0.2
    STATES_IN_OBS = ['dist_to_goal', 'rel_goal', 'radius', 'heading_ego_frame', 'pref_speed', 'other_agents_states']

    @classmethod
    def state_info(cls):
        """name -> (shape, (low, high)); the in-scope keys of STATE_INFO_DICT (config.py:104-215)."""
        k = cls.MAX_NUM_AGENTS_IN_ENVIRONMENT - 1
        inf = np.inf
        return {'dist_to_goal': ((1,), (-inf, inf)), 'radius': ((1,), (0, inf)), 'rel_goal': ((2,), (-inf, inf)),
                'heading_ego_frame': ((1,), (-np.pi, np.pi)), 'heading_global_frame': ((1,), (-np.pi, np.pi)),
                'pos_global_frame': ((2,), (-inf, inf)), 'pref_speed': ((1,), (0, inf)),
                'num_other_agents': ((1,), (0, inf)), 'other_agent_states': ((10,), (-inf, inf)),
                'other_agents_states': ((k, 10), (-inf, inf)), 'laserscan': ((cls.LASERSCAN_LENGTH,), (0., 6.)),
                'use_ppo': ((1,), (0., 1.)), 'local_grid': ((60, 60), (0., 1.))}


class Box(object):
    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype), dtype
        self.shape = self.low.shape

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)


class Dict(object):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})


class Agent(object):
    """Scenario record with the reference constructor's argument order (agent.py:9-10); after reset the
    env replaces `env.agents` entries by live views whose attributes read the device state."""

    def __init__(self, start_x, start_y, goal_x, goal_y, radius, pref_speed, initial_heading, policy,
                 dynamics_model, sensors, id, cooperation_coef=1.0):
        self.start = (float(start_x), float(start_y))
        self.goal_global_frame = np.array([goal_x, goal_y], dtype=np.float64)
        self.pos_global_frame = np.array([start_x, start_y], dtype=np.float64)
        self.radius, self.pref_speed = float(radius), float(pref_speed)
        self.initial_heading = initial_heading
        self.policy = policy() if isinstance(policy, type) else policy
        self.dynamics_model = dynamics_model() if isinstance(dynamics_model, type) else dynamics_model
        self.sensors = [s() if isinstance(s, type) else s for s in sensors]
        self.id = id
        self.cooperation_coef = float(cooperation_coef)


class AgentView(object):
    """Read-only live view of one agent slot (attribute names of agent.py:9-109)."""

    def __init__(self, env, index, spec):
        self._env, self._i, self._spec = env, index, spec
        self.id, self.policy, self.sensors = spec.id, spec.policy, spec.sensors
        self.radius, self.pref_speed = spec.radius, spec.pref_speed
        self.goal_global_frame = spec.goal_global_frame
        self.cooperation_coef = spec.cooperation_coef

    def _s(self, name):
        return float(self._env._snapshot()[name][self._i])

    def _flag(self, bit):
        return bool(int(self._env._snapshot()["status"][self._i]) & bit)

    pos_global_frame = property(lambda s: np.array([s._s("pos_x"), s._s("pos_y")]))
    vel_global_frame = property(lambda s: np.array([s._s("vel_x"), s._s("vel_y")]))
    heading_global_frame = property(lambda s: s._s("heading"))
    heading_ego_frame = property(lambda s: s._s("heading_ego"))
    speed_global_frame = property(lambda s: s._s("speed"))
    delta_heading_global_frame = property(lambda s: s._s("delta_heading"))
    dist_to_goal = property(lambda s: s._s("dist_to_goal"))
    rel_goal = property(lambda s: s.goal_global_frame - s.pos_global_frame)
    time_remaining_to_reach_goal = property(lambda s: s._s("time_remaining"))
    t = property(lambda s: s._s("t"))
    step_num = property(lambda s: int(s._env._snapshot()["step_num"][s._i]))
    is_at_goal = property(lambda s: s._flag(1))
    in_collision = property(lambda s: s._flag(2))
    ran_out_of_time = property(lambda s: s._flag(4))
    is_done = property(lambda s: s._flag(8))
    was_at_goal_already = property(lambda s: s._flag(16))
    was_in_collision_already = property(lambda s: s._flag(32))

    @property
    def straight_line_time_to_reach_goal(self):
        sp = self._spec
        return (np.linalg.norm(np.array(sp.start) - sp.goal_global_frame) - Config.NEAR_GOAL_THRESHOLD) / sp.pref_speed


class _FrozenAgent(object):
    """prev_episode_agents entry: a value copy of the final AgentView attributes (env.py:404-405)."""
    FIELDS = ["id", "radius", "pref_speed", "pos_global_frame", "vel_global_frame", "heading_global_frame",
              "dist_to_goal", "t", "step_num", "is_at_goal", "in_collision", "ran_out_of_time", "is_done",
              "straight_line_time_to_reach_goal", "time_remaining_to_reach_goal", "goal_global_frame"]

    def __init__(self, view):
        for f in self.FIELDS:
            setattr(self, f, copy.copy(getattr(view, f)))


class CollisionAvoidanceEnv(object):
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 30}

    def __init__(self, device="cuda:0"):
        self.id = 0
        self.device = device
        self.dt_nominal = Config.DT
        self.num_agents = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT
        self.max_heading_change, self.min_heading_change, self.min_speed, self.max_speed = 4, -4, -4, 4
        self.action_space = Box([self.min_speed, self.min_heading_change], [self.max_speed, self.max_heading_change])
        info = Config.state_info()
        self.observation_space = Dict({s: Box(info[s][1][0] * np.ones(info[s][0]), info[s][1][1] * np.ones(info[s][0]))
                                       for s in Config.STATES_IN_OBS})
        self.agents = None
        self.default_agents = None
        self.default_obstacles = []
        self.prev_episode_agents = None
        self.episode_number = 0
        self.episode_step_number = 0
        self.total_number_of_steps = 0
        self.test_case_index = 0
        self.plot_save_dir = None
        self.plot_policy_name = None
        self.perturbed_obs = None
        self._benv = None
        self._snap = None
        self._sig = None

    # -- setters of the reference surface ---------------------------------------------------------
    def set_agents(self, agents):
        if isinstance(agents, tuple):  # scenario functions of test_cases.py return (agents, obstacles)
            agents, self.default_obstacles = agents
        self.default_agents = agents

    def set_obstacles(self, obstacles):
        """obstacles: list of 4-corner polygons [(xu,yu),(xl,yu),(xl,yl),(xu,yl)] (test_cases.py:2496)
        or [xl, yl, xu, yu] rows."""
        self.default_obstacles = list(obstacles)

    def set_static_map(self, map_filename):
        self.static_map_filename = map_filename

    def set_plot_save_dir(self, plot_save_dir):
        self.plot_save_dir = plot_save_dir  # plotting is out of scope

    def set_perturbed_info(self, perturbed_obs):
        self.perturbed_obs = perturbed_obs

    def close(self):
        if self._benv is not None:
            self._benv.close()
            self._benv = None

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _rects(obstacles):
        out = []
        for o in obstacles:
            o = np.asarray(o, dtype=np.float64)
            if o.shape == (4, 2):
                out.append([o[1, 0], o[3, 1], o[3, 0], o[1, 1]])  # xl, yl, xu, yu
            else:
                out.append(list(o.reshape(4)))
        return np.asarray(out, dtype=np.float64).reshape(-1, 4)

    def _game_over_mode(self):
        if Config.EVALUATE_MODE:
            return "all" if Config.HOMOGENEOUS_TESTING else "agent0"
        return "agent0" if Config.TRAIN_SINGLE_AGENT else "learning"

    def _snapshot(self):
        if self._snap is None:
            import torch
            torch.cuda.synchronize()
            st = self._benv.state()
            self._snap = {k: st[k][0].cpu().numpy() for k in
                          ("pos_x", "pos_y", "vel_x", "vel_y", "heading", "heading_ego", "speed", "delta_heading",
                           "dist_to_goal", "time_remaining", "t", "status", "step_num")}
        return self._snap

    def _obs_dict(self):
        import torch
        torch.cuda.synchronize()
        b = self._benv
        ego = b.obs_ego[0].cpu().numpy().astype(np.float64)
        oas = b.obs_oas[0].cpu().numpy().astype(np.float64)
        laser = b.obs_laser[0].cpu().numpy().astype(np.float64) if b.laserscan else None
        grid = None
        if 'local_grid' in Config.STATES_IN_OBS and b.Kobs > 0:  # OccupancyGridSensor (parity unpinned: cv2 restated)
            grid = b.sense_occupancy_grid()[0].cpu().numpy().astype(bool)
        info = Config.state_info()
        obs = {}
        for i in range(Config.MAX_NUM_AGENTS_IN_ENVIRONMENT):
            d = {}
            live = i < len(self.agents)
            for s in Config.STATES_IN_OBS:
                if not live:
                    d[s] = np.zeros(info[s][0], dtype=np.float32)
                elif s == 'dist_to_goal':
                    d[s] = np.array(ego[i, 0])
                elif s == 'rel_goal':
                    d[s] = ego[i, 1:3].copy()
                elif s == 'radius':
                    d[s] = np.array(ego[i, 3])
                elif s == 'heading_ego_frame':
                    d[s] = np.array(ego[i, 4])
                elif s == 'heading_global_frame':
                    d[s] = np.array(ego[i, 5])
                elif s == 'pos_global_frame':
                    d[s] = ego[i, 6:8].copy()
                elif s == 'pref_speed':
                    d[s] = np.array(ego[i, 8])
                elif s == 'num_other_agents':
                    d[s] = np.array(ego[i, 9])
                elif s == 'use_ppo':
                    d[s] = np.array(bool(ego[i, 10]))
                elif s == 'other_agents_states':
                    d[s] = oas[i].copy()
                elif s == 'other_agent_states':
                    d[s] = oas[i, 0].copy()
                elif s == 'laserscan':
                    d[s] = laser[i].copy() if laser is not None else np.zeros(16)
                elif s == 'local_grid':
                    d[s] = grid[i].copy() if grid is not None else np.zeros((60, 60), dtype=bool)
                else:
                    raise KeyError("observation key %r is outside the hot-path scope (SURVEY.md 8(a))" % s)
            obs[i] = d
        return obs

    # -- reset / step -----------------------------------------------------------------------------
    def reset(self):
        if self.default_agents is None:
            raise RuntimeError("call set_agents() before reset() (deviation D1: set_agents is honoured)")
        if self.agents is not None:
            self.prev_episode_agents = [_FrozenAgent(a) for a in self.agents]
        specs = list(self.default_agents)
        M = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT
        n = len(specs)
        if n > M:
            raise ValueError("more agents than Config.MAX_NUM_AGENTS_IN_ENVIRONMENT")
        rects = self._rects(self.default_obstacles)
        laser = any(isinstance(s, LaserScanSensor) for a in specs for s in a.sensors)
        sig = (M, max(len(rects), 0), laser, self._game_over_mode(), Config.COLLISION_AV_W_STATIC_AGENT, Config.DT)
        if self._benv is None or sig != self._sig:
            self.close()
            from .batched_env import BatchedCollisionAvoidanceEnv
            self._benv = BatchedCollisionAvoidanceEnv(1, M, max_obstacles=len(rects), game_over_mode=sig[3],
                                                      collide_with_static=sig[4], laserscan=laser, device=self.device,
                                                      dt=Config.DT)
            self._sig = sig
        a6 = np.zeros((1, M, 6))
        a6[0, :, 4] = 1.0
        a6[0, :, 5] = 0.1
        h0 = np.zeros((1, M))
        pol = np.zeros((1, M), dtype=np.int32)
        dyn = np.zeros((1, M), dtype=np.int32)
        coop = np.ones((1, M))
        for i, a in enumerate(specs):
            a6[0, i] = [a.start[0], a.start[1], a.goal_global_frame[0], a.goal_global_frame[1], a.pref_speed, a.radius]
            h0[0, i] = (np.arctan2(a.goal_global_frame[1] - a.start[1], a.goal_global_frame[0] - a.start[0])
                        if a.initial_heading is None else a.initial_heading)
            pol[0, i] = a.policy.policy_id
            dyn[0, i] = a.dynamics_model.dynamics_id
            coop[0, i] = a.cooperation_coef
        self._benv.set_scenarios(a6, pol, dyn, heading0=h0, n_agents=[n], coop=coop,
                                 obstacles=rects[None] if len(rects) else None,
                                 n_obst=[len(rects)] if len(rects) else None)
        self._benv.reset()
        self.episode_number += 1
        self.episode_step_number = 0
        self._snap = None
        self.agents = [AgentView(self, i, a) for i, a in enumerate(specs)]
        return self._obs_dict()

    def step(self, actions, dt=None):
        if self._benv is None:
            raise RuntimeError("reset() first")
        if dt is not None and dt != self._sig[5]:
            raise ValueError("per-call dt differs from Config.DT fixed at reset")
        M = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT
        ext = np.zeros((1, M, 2), dtype=np.float32)
        if actions is not None:
            for i, a in enumerate(self.agents):
                try:
                    v = actions[i]
                except (KeyError, IndexError, TypeError):
                    continue
                if v is None:
                    continue
                v = np.atleast_1d(np.asarray(v, dtype=np.float32)).ravel()
                ext[0, i, :min(2, v.size)] = v[:2]
        self.episode_step_number += 1
        self.total_number_of_steps += 1
        _, rew, go, _ = self._benv.step(ext)
        self._snap = None
        obs = self._obs_dict()
        rewards = rew[0, :len(self.agents)].double().cpu().numpy()
        if Config.TRAIN_SINGLE_AGENT:
            rewards = rewards[0]
        game_over = bool(go[0].item())
        which = {a.id: bool(a.is_done) for a in self.agents}
        return obs, rewards, game_over, {'which_agents_done': which}


def get_testcase_two_agents(policies=(LearningPolicy, NonCooperativePolicy)):
    """Two agents swapping corners of a 6 m square, heading 0.5 rad: the scenario shape of
    experiments/src/example.py's tc.get_testcase_two_agents() (policy of agent 1 is a parameter because
    GA3C-CADRL inference is supplied separately)."""
    return [Agent(-3, -3, 3, 3, 0.5, 1.0, 0.5, policies[0], UnicycleDynamics, [OtherAgentsStatesSensor], 0),
            Agent(3, 3, -3, -3, 0.5, 1.0, 0.5, policies[1], UnicycleDynamics, [OtherAgentsStatesSensor], 1)]
