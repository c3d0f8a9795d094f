"""BatchedCollisionAvoidanceEnv: N worlds x M agents of the reference's CollisionAvoidanceEnv
(gym_collision_avoidance/envs/collision_avoidance_env.py) stepped by hand-written HIP kernels.

Vectorised convention of the reference's DummyVecEnv use (experiments/src/env_utils.py:29-31,
envs/wrappers.py:101-106): step(actions[N, M, 2]) -> (obs, rewards[N, M], game_over[N], info);
observations and outputs are PyTorch-ROCm tensors that the kernels write in place (zero-copy).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (device index) -> hipStream_t as an int

_TORCH_DT = {"f8": torch.float64, "f4": torch.float32, "u4": torch.int32, "i4": torch.int32}


class _DevArray(object):
    """Minimal __cuda_array_interface__ holder: zero-copy torch view of a raw device pointer."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<" + typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


GAME_OVER_MODES = {"agent0": _lib.GO_AGENT0, "all": _lib.GO_ALL, "learning": _lib.GO_LEARNING}


class BatchedCollisionAvoidanceEnv(object):
    def __init__(self, n_worlds, max_agents=10, n_scenarios=None, max_obstacles=0, game_over_mode="agent0",
                 collide_with_static=False, laserscan=False, device="cuda:0", dt=0.1, rvo_max_neighbors=0):
        self.L = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedCollisionAvoidanceEnv runs on a ROCm device only (got %s)" % device)
        self.N, self.M, self.K = int(n_worlds), int(max_agents), int(max_agents) - 1
        self.S = int(n_scenarios) if n_scenarios else self.N
        self.Kobs = int(max_obstacles)
        self.laserscan = bool(laserscan)
        if isinstance(game_over_mode, str):
            game_over_mode = GAME_OVER_MODES[game_over_mode]
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dev_index = int(idx)
        self.cfg = _lib.CagymConfig(self.N, self.M, self.S, self.Kobs, int(game_over_mode),
                                    int(bool(collide_with_static)), int(self.laserscan), int(idx), float(dt),
                                    int(rvo_max_neighbors), 0)  # rvo_max_neighbors 0 = max_agents (RVOPolicy.py:15)
        self.h = C.c_void_p()
        _lib.check(self.L, None, self.L.cagym_create(C.byref(self.cfg), C.byref(self.h)), "cagym_create")
        N, M, K = self.N, self.M, self.K
        dev = self.device
        self.obs_oas = torch.zeros((N, M, K, 10), dtype=torch.float32, device=dev)
        self.obs_ego = torch.zeros((N, M, _lib.EGO_WIDTH), dtype=torch.float32, device=dev)
        self.obs_laser = torch.zeros((N, M, 16), dtype=torch.float32, device=dev) if self.laserscan else None
        self.reward = torch.zeros((N, M), dtype=torch.float32, device=dev)
        self.flags = torch.zeros((N, M), dtype=torch.uint8, device=dev)
        self.game_over = torch.zeros((N,), dtype=torch.uint8, device=dev)
        self._out = self._outputs(self.obs_oas, self.obs_ego, self.obs_laser, self.reward, self.flags, self.game_over)
        self._state = None
        self._side = None  # side stream of step_overlapped (created on first use)

    # ---- plumbing ------------------------------------------------------------------------------
    @staticmethod
    def _outputs(oas, ego, laser, reward, flags, go):
        p = lambda t: None if t is None else t.data_ptr()
        return _lib.CagymOutputs(p(oas), p(ego), p(laser), p(reward), p(flags), p(go))

    def _stream(self):
        # torch's current stream of the handle's device (the raw-handle accessor costs 0.3 us, the Stream object 2 us per launch)
        if _raw_stream is not None:
            return C.c_void_p(_raw_stream(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize(self.device)
            self.L.cagym_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scenarios (set_agents / _init_agents / _init_static_map) ---------------------------------
    def set_scenarios(self, agents6, policy_id, dynamics_id, heading0=None, n_agents=None, coop=None,
                      obstacles=None, n_obst=None):
        S, M = self.S, self.M
        a6 = np.ascontiguousarray(np.asarray(agents6, dtype=np.float64).reshape(S, M, 6))
        pol = np.ascontiguousarray(np.broadcast_to(np.asarray(policy_id, dtype=np.int32), (S, M)))
        dyn = np.ascontiguousarray(np.broadcast_to(np.asarray(dynamics_id, dtype=np.int32), (S, M)))
        opt = lambda x, dt, shape: None if x is None else np.ascontiguousarray(np.asarray(x, dtype=dt).reshape(shape))
        h0 = opt(heading0, np.float64, (S, M))
        na = opt(n_agents, np.int32, (S,))
        co = opt(coop, np.float64, (S, M))
        ob = no = None
        if obstacles is not None and self.Kobs:
            o = np.asarray(obstacles, dtype=np.float64).reshape(S, -1, 4)
            ob = np.zeros((S, self.Kobs, 4), dtype=np.float64)
            ob[:, :o.shape[1]] = o
            no = np.ascontiguousarray(np.asarray(n_obst, dtype=np.int32).reshape(S))
        p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
        with torch.cuda.device(self.device):
            rc = self.L.cagym_set_scenarios(self.h, p(a6), p(h0), p(pol), p(dyn), p(na), p(co), p(ob), p(no),
                                            self._stream())
        _lib.check(self.L, self.h, rc, "cagym_set_scenarios")

    def sense_occupancy_grid(self, out=None):
        """'local_grid' observation of every agent (OccupancyGridSensor.sense): uint8 [N, M, 60, 60], 1 = occupied.
        Parity unpinned (cv2.warpAffine is restated, OpenCV is not installed): see include/cagym.h."""
        if out is None:
            out = torch.empty((self.N, self.M, 60, 60), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.L.cagym_occupancy_grid(self.h, out.data_ptr(), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_occupancy_grid")
        return out

    def generate_scenarios(self, seed, n_agents=None, ego_policy=5, ego_dynamics=0, other_policies=(5, 1), p_b=0.5,
                           other_dynamics=0, side=7.5, min_travel=4.0, min_sep=1.5, radius=0.5, pref_speed=1.0,
                           coop=0.5, max_tries=100000, check=True):
        """Fill the scenario pool ON DEVICE with the rule of train_agents_random_positions (test_cases.py:1362-1463):
        no host sampling, no upload.  n_agents = int or (min, max); agent 0 gets (ego_policy, ego_dynamics), every
        other agent other_policies[1] with probability p_b else other_policies[0].  Returns the number of agents
        whose rejection loop hit max_tries (0 in any sane configuration) when check=True."""
        if n_agents is None:
            n_agents = self.M
        n_min, n_max = (n_agents, n_agents) if np.isscalar(n_agents) else n_agents
        P = _lib.CagymGenParams(int(seed), int(n_min), int(n_max), int(ego_policy), int(ego_dynamics),
                                int(other_policies[0]), int(other_policies[1]), int(other_dynamics), int(max_tries),
                                float(p_b), float(side), float(min_travel), float(min_sep), float(radius),
                                float(pref_speed), float(coop))
        nf = C.c_int32(0)
        with torch.cuda.device(self.device):
            rc = self.L.cagym_generate_scenarios(self.h, C.byref(P), C.byref(nf) if check else None, self._stream())
        _lib.check(self.L, self.h, rc, "cagym_generate_scenarios")
        return int(nf.value) if check else None

    def scenarios(self):
        """Zero-copy device views of the scenario pool: agents6 [S,M,6], policy / dynamics [S,M], n_agents [S], coop."""
        sp = _lib.CagymScenarioPtrs()
        _lib.check(self.L, self.h, self.L.cagym_get_scenarios(self.h, C.byref(sp)), "cagym_get_scenarios")
        S, M = self.S, self.M
        spec = {"agents6": ((S, M, 6), "f8"), "policy": ((S, M), "i4"), "dynamics": ((S, M), "i4"),
                "n_agents": ((S,), "i4"), "coop": ((S, M), "f8")}
        return {k: torch.as_tensor(_DevArray(getattr(sp, k), shp, ts), device=self.device) for k, (shp, ts) in spec.items()}

    # ---- gym surface ----------------------------------------------------------------------------
    def _obs(self):
        obs = {"other_agents_states": self.obs_oas, "ego": self.obs_ego}
        if self.laserscan:
            obs["laserscan"] = self.obs_laser
        return obs

    def reset(self, world_mask=None, advance_episode=False):
        m = None
        if world_mask is not None:
            m = torch.as_tensor(world_mask, device=self.device).to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            rc = self.L.cagym_reset(self.h, None if m is None else m.data_ptr(), int(bool(advance_episode)),
                                    C.byref(self._out), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_reset")
        return self._obs()

    def step(self, actions=None, auto_reset=False):
        """One env.step() of every world.  auto_reset=True: finished worlds restart inside the same launch
        (VecEnv semantics: the returned observation is the first one of the new episode)."""
        a = None
        if actions is not None:
            a = torch.as_tensor(actions, device=self.device).to(torch.float32).reshape(self.N, self.M, 2).contiguous()
        fn = self.L.cagym_step_autoreset if auto_reset else self.L.cagym_step
        rc = fn(self.h, None if a is None else a.data_ptr(), C.byref(self._out), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_step")
        return self._obs(), self.reward, self.game_over, {"flags": self.flags}

    # ---- the split step: env.step() in two launches (include/cagym.h: cagym_step_begin / cagym_step_finish) ------------------
    def step_begin(self, stream=None):
        """First half of step(): the internal RVO policies' half-planes and linear programs on the current state (they do not
        depend on the external actions: env.py:287-340 gathers every agent's action before any agent moves).  `stream`: a
        torch.cuda.Stream to run it on BESIDE the producer of the external actions (default: the current stream); the caller
        orders step_finish behind it (step_overlapped does)."""
        raw = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.L, self.h, self.L.cagym_step_begin(self.h, raw), "cagym_step_begin")

    def step_finish(self, actions=None, auto_reset=False):
        """Second half of step(): everything else, with every agent's action in hand.  Same results as step(), bit for bit."""
        a = None
        if actions is not None:
            a = torch.as_tensor(actions, device=self.device).to(torch.float32).reshape(self.N, self.M, 2).contiguous()
        rc = self.L.cagym_step_finish(self.h, None if a is None else a.data_ptr(), C.byref(self._out), int(bool(auto_reset)), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_step_finish")
        return self._obs(), self.reward, self.game_over, {"flags": self.flags}

    def step_overlapped(self, policy, actions, auto_reset=False):
        """step() with a device policy in the loop: `policy(actions)` fills the external actions on the current stream (e.g.
        GA3CCADRLPolicy.act) WHILE the RVO half of the step runs on a side stream; the rest of the step follows both.
        Worth it when the policy leaves the GPU idle (a host-side or remote policy).  NOT for cagym_ga3c_act on the same GPU:
        measured on MI355X (profiles/r4/cfg4_overlap_trace_*.txt) the two kernels do run side by side, but sharing the CUs halves
        each one's occupancy and both take twice as long - 0.251 ms per cfg4 step against 0.194 ms for the fused launch."""
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(main)          # the previous step's state
        self.step_begin(stream=self._side)
        policy(actions)
        main.wait_stream(self._side)
        return self.step_finish(actions, auto_reset=auto_reset)

    def alloc_rollout(self, n_steps, obs=True):
        T, N, M, K, dev = int(n_steps), self.N, self.M, self.K, self.device
        buf = {"reward": torch.empty((T, N, M), dtype=torch.float32, device=dev),
               "flags": torch.empty((T, N, M), dtype=torch.uint8, device=dev),
               "game_over": torch.empty((T, N), dtype=torch.uint8, device=dev)}
        if obs:
            buf["other_agents_states"] = torch.empty((T, N, M, K, 10), dtype=torch.float32, device=dev)
            buf["ego"] = torch.empty((T, N, M, _lib.EGO_WIDTH), dtype=torch.float32, device=dev)
            if self.laserscan:
                buf["laserscan"] = torch.empty((T, N, M, 16), dtype=torch.float32, device=dev)
        return buf

    def rollout(self, n_steps, auto_reset=True, out=None):
        """n_steps env steps in one launch (all agents internally driven); returns [T, ...] buffers."""
        if out is None:
            out = self.alloc_rollout(n_steps)
        o = self._outputs(out.get("other_agents_states"), out.get("ego"), out.get("laserscan"), out.get("reward"),
                          out.get("flags"), out.get("game_over"))
        # (no torch.cuda.device context here and in step(): the library switches to the handle's device itself - DEVGUARD)
        rc = self.L.cagym_rollout(self.h, int(n_steps), int(bool(auto_reset)), C.byref(o), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_rollout")
        return out

    def kernel_name(self, rollout=True, auto_reset=True):
        """The kernel instantiation the library launches for this handle (as rocprofv3 --kernel-trace names it)."""
        buf = C.create_string_buffer(128)
        _lib.check(self.L, self.h, self.L.cagym_kernel_name(self.h, int(bool(rollout)), int(bool(auto_reset)), buf, 128),
                   "cagym_kernel_name")
        return buf.value.decode()

    def sense_laserscan(self, out=None):
        if out is None:
            out = torch.empty((self.N, self.M, 16), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.L.cagym_laserscan(self.h, out.data_ptr(), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_laserscan")
        return out

    # ---- zero-copy state views --------------------------------------------------------------------
    def state(self):
        if self._state is None:
            sp = _lib.CagymStatePtrs()
            _lib.check(self.L, self.h, self.L.cagym_get_state(self.h, C.byref(sp)), "cagym_get_state")
            N, M = self.N, self.M
            shapes = {"action": (N, M, 2), "n_agents": (N,), "episode": (N,), "stat_return": (N,),
                      "stat_episodes": (N,), "stat_steps": (N,), "stat_outcomes": (N, 3),
                      "map_bits": (self.S, 300, 10)}
            st = {}
            for name, ts in _lib.STATE_FIELDS:
                ptr = getattr(sp, name)
                if not ptr:
                    continue
                shape = shapes.get(name, (N, M))
                t = torch.as_tensor(_DevArray(ptr, shape, ts), device=self.device)
                st[name] = t.view(torch.int32) if ts == "u4" else t  # uint32 has few torch kernels
            self._state = st
        return self._state

    def episode_stats(self):
        s = self.state()
        return {k: s[k] for k in ("stat_return", "stat_episodes", "stat_steps", "stat_outcomes")}

    def packed_episode_stats(self, out=None):
        """[N, 6] int32 records (stats.py layout) written by ONE kernel on the current stream (cagym_pack_episode_stats)."""
        if out is None:
            out = torch.empty((self.N, 6), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.L.cagym_pack_episode_stats(self.h, out.data_ptr(), self._stream())
        _lib.check(self.L, self.h, rc, "cagym_pack_episode_stats")
        return out

    # ---- parity-test interface (f/u/i accessors used by tests/golden_util.replay) ----------------------------
    def f(self, name):
        torch.cuda.synchronize(self.device)
        s = self.state()
        g = lambda k: s[k].cpu().numpy()
        if name == "pos":
            return np.stack([g("pos_x"), g("pos_y")], -1)
        if name == "vel":
            return np.stack([g("vel_x"), g("vel_y")], -1)
        if name == "rel_goal":
            return np.stack([g("goal_x") - g("pos_x"), g("goal_y") - g("pos_y")], -1)
        if name == "oas":
            return self.obs_oas.double().cpu().numpy()
        if name == "laserscan":
            return self.obs_laser.double().cpu().numpy()
        if name == "reward":
            return self.reward.double().cpu().numpy()
        if name == "action":
            return g("action").astype(np.float64)
        return g(name)

    def u(self, name):
        torch.cuda.synchronize(self.device)
        if name == "game_over":
            return self.game_over.cpu().numpy()
        st = self.state()["status"].cpu().numpy().astype(np.uint32)
        bit = {"is_at_goal": _lib.FLAG_AT_GOAL, "was_at_goal_already": _lib.FLAG_WAS_AT_GOAL,
               "in_collision": _lib.FLAG_IN_COLLISION, "was_in_collision_already": _lib.FLAG_WAS_IN_COLLISION,
               "ran_out_of_time": _lib.FLAG_RAN_OUT_OF_TIME, "is_done": _lib.FLAG_DONE}[name]
        return ((st & bit) != 0).astype(np.uint8)

    def i(self, name):
        torch.cuda.synchronize(self.device)
        key = {"step_num": "step_num", "num_other_agents_observed": "n_observed"}[name]
        return self.state()[key].cpu().numpy()
