"""ctypes front-end of the parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcagym_oracle.so")

F = dict(pos=0, vel=1, heading=2, speed=3, delta_heading=4, dist_to_goal=5, past_dist_to_goal=6,
         heading_ego=7, vel_ego=8, ref_prll=9, rel_goal=10, time_remaining=11, t=12, past_actions=13,
         reward=14, oas=15, laserscan=16, action=17)
U = dict(is_at_goal=0, was_at_goal_already=1, in_collision=2, was_in_collision_already=3,
         ran_out_of_time=4, is_done=5, game_over=6, map=7)
I = dict(step_num=0, num_other_agents_observed=1)
F_WIDTH = dict(pos=2, vel=2, heading=1, speed=1, delta_heading=1, dist_to_goal=1, past_dist_to_goal=1,
               heading_ego=1, vel_ego=2, ref_prll=2, rel_goal=2, time_remaining=1, t=1, past_actions=4,
               reward=1, laserscan=16, action=2)
GO_AGENT0, GO_ALL, GO_LEARNING = 0, 1, 2


class Config(C.Structure):
    _fields_ = [("n_worlds", C.c_int32), ("max_agents", C.c_int32), ("max_obstacles", C.c_int32),
                ("game_over_mode", C.c_int32), ("collide_with_static", C.c_int32), ("laserscan", C.c_int32),
                ("dt", C.c_double), ("rvo_max_neighbors", C.c_int32), ("reserved", C.c_int32)]


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("cagym_oracle.c", "cagym_oracle_ig.c", "cagym_oracle_gen.c", "cagym_oracle_grid.c", "cagym_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, "libcagym_oracle.so"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()  # no-op unless a source is newer than the library
        L = C.CDLL(LIB)
        L.cao_create.restype = C.c_void_p
        L.cao_create.argtypes = [C.POINTER(Config)]
        L.cao_destroy.argtypes = [C.c_void_p]
        L.cao_set_scenario.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.cao_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.cao_step.argtypes = [C.c_void_p, C.c_void_p]
        L.cao_run.argtypes = [C.c_void_p, C.c_int]
        L.cao_set_threads.restype = C.c_int
        L.cao_set_threads.argtypes = [C.c_int]
        for n, t in (("cao_f64", C.c_double), ("cao_u8", C.c_uint8), ("cao_i32", C.c_int32)):
            getattr(L, n).restype = C.POINTER(t)
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
        L.cao_rasterize.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cao_orca_action.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_double] * 3 + [C.c_void_p]
        L.cao_orca_action_ex.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_double] * 3 + [C.c_int, C.c_void_p, C.c_int]
                                         + [C.c_void_p] * 4)
        L.cao_ga3c_states.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cao_edt.argtypes = [C.c_void_p] * 3
        L.cao_ig_check_visibility.argtypes = [C.c_void_p] * 3
        L.cao_ig_visible.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.cao_ig_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_double, C.c_double, C.c_void_p]
        L.cao_ig_reward.restype = C.c_double
        L.cao_ig_reward.argtypes = [C.c_void_p, C.c_void_p]
        L.cao_ig_next_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.cao_ig_rand_primitive.restype = C.c_uint32
        L.cao_ig_rand_primitive.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.cao_ig_rollout.restype = C.c_double
        L.cao_ig_rollout.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                        C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                                        C.c_void_p, C.c_void_p]
        L.cao_occupancy_grid.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.cao_generate_scenarios.restype = C.c_int
        L.cao_generate_scenarios.argtypes = [C.c_uint64] + [C.c_int] * 10 + [C.c_double] * 7 + [C.c_void_p] * 5
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleEnv(object):
    """N worlds x M agent slots, fp64, scalar CPU."""

    def __init__(self, N, M, max_obstacles=0, game_over_mode=GO_AGENT0, collide_with_static=False,
                 laserscan=False, dt=0.1, rvo_max_neighbors=0):
        self.N, self.M, self.K, self.Kobs = N, M, M - 1, max_obstacles
        self.cfg = Config(N, M, max_obstacles, game_over_mode, int(collide_with_static), int(laserscan), dt,
                          int(rvo_max_neighbors), 0)
        self.L = lib()
        self.h = self.L.cao_create(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.cao_destroy(self.h)
            self.h = None

    def set_scenario(self, agents6, policy_id, dynamics_id, heading0=None, n_agents=None, coop=None,
                     obstacles=None, n_obst=None):
        N, M = self.N, self.M
        a6 = np.ascontiguousarray(np.asarray(agents6, dtype=np.float64).reshape(N, M, 6))
        pol = np.ascontiguousarray(np.broadcast_to(np.asarray(policy_id, dtype=np.int32), (N, M)))
        dyn = np.ascontiguousarray(np.broadcast_to(np.asarray(dynamics_id, dtype=np.int32), (N, M)))
        h0 = None if heading0 is None else np.ascontiguousarray(np.asarray(heading0, dtype=np.float64).reshape(N, M))
        na = None if n_agents is None else np.ascontiguousarray(np.asarray(n_agents, dtype=np.int32).reshape(N))
        co = None if coop is None else np.ascontiguousarray(np.asarray(coop, dtype=np.float64).reshape(N, M))
        ob = no = None
        if obstacles is not None and self.Kobs:
            ob = np.zeros((N, self.Kobs, 4), dtype=np.float64)
            o = np.asarray(obstacles, dtype=np.float64)
            ob[:, :o.shape[-2], :] = o.reshape(N, -1, 4)
            no = np.ascontiguousarray(np.asarray(n_obst, dtype=np.int32).reshape(N))
        self.L.cao_set_scenario(self.h, _p(a6), _p(h0), _p(pol), _p(dyn), _p(na), _p(co), _p(ob), _p(no))

    def reset(self, world_mask=None):
        m = None if world_mask is None else np.ascontiguousarray(np.asarray(world_mask, dtype=np.uint8))
        self.L.cao_reset(self.h, _p(m))

    def step(self, ext_actions=None):
        a = None if ext_actions is None else np.ascontiguousarray(
            np.asarray(ext_actions, dtype=np.float64).reshape(self.N, self.M, 2))
        self.L.cao_step(self.h, _p(a))

    def run(self, n_steps):
        """n_steps x (step every world, restart the finished ones on the same scenario) without leaving C."""
        self.L.cao_run(self.h, int(n_steps))

    def f(self, name):
        w = self.K * 10 if name == "oas" else F_WIDTH[name]
        arr = np.ctypeslib.as_array(self.L.cao_f64(self.h, F[name]), shape=(self.N * self.M * w,))
        if name == "oas":
            return arr.reshape(self.N, self.M, self.K, 10)
        if name == "past_actions":
            return arr.reshape(self.N, self.M, 2, 2)
        return arr.reshape(self.N, self.M, w) if w > 1 else arr.reshape(self.N, self.M)

    def u(self, name):
        if name == "game_over":
            return np.ctypeslib.as_array(self.L.cao_u8(self.h, U[name]), shape=(self.N,))
        if name == "map":
            return np.ctypeslib.as_array(self.L.cao_u8(self.h, U[name]), shape=(self.N, 300, 300))
        return np.ctypeslib.as_array(self.L.cao_u8(self.h, U[name]), shape=(self.N, self.M))

    def i(self, name):
        return np.ctypeslib.as_array(self.L.cao_i32(self.h, I[name]), shape=(self.N, self.M))

    def ga3c_states(self, max_observed=None):
        out = np.zeros((self.N, self.M, 76))
        self.L.cao_ga3c_states(self.h, self.K if max_observed is None else int(max_observed), _p(out))
        return out


def rasterize(obstacles):
    o = np.ascontiguousarray(np.asarray(obstacles, dtype=np.float64).reshape(-1, 4))
    out = np.zeros((300, 300), dtype=np.uint8)
    lib().cao_rasterize(_p(o), o.shape[0], _p(out))
    return out


def orca_action(pos, vel, goal, pref_speed, radius, ego, heading, collab=0.5, dt=0.1):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    vel = np.ascontiguousarray(vel, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    ps = np.ascontiguousarray(pref_speed, dtype=np.float64)
    rd = np.ascontiguousarray(radius, dtype=np.float64)
    out = np.zeros(2)
    lib().cao_orca_action(pos.shape[0], ego, _p(pos), _p(vel), _p(goal), _p(ps), _p(rd), float(heading),
                          float(collab), float(dt), _p(out))
    return out


def orca_action_ex(pos, vel, goal, pref_speed, radius, ego, heading, collab=0.5, dt=0.1, max_neighbors=10, rects=None):
    """RVO ego solve with static rectangles [n, 4] = xl, yl, xu, yu.  Returns dict(action[2], new_vel[2] fp32,
    lines [n_lines, 4] (point, direction), n_obst_lines)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    vel = np.ascontiguousarray(vel, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    ps = np.ascontiguousarray(pref_speed, dtype=np.float64)
    rd = np.ascontiguousarray(radius, dtype=np.float64)
    rc = None if rects is None else np.ascontiguousarray(np.asarray(rects, dtype=np.float64).reshape(-1, 4))
    out = np.zeros(2)
    nv = np.zeros(2, dtype=np.float32)
    lines = np.zeros((192, 4), dtype=np.float32)
    nl = np.zeros(2, dtype=np.int32)
    lib().cao_orca_action_ex(pos.shape[0], ego, _p(pos), _p(vel), _p(goal), _p(ps), _p(rd), float(heading), float(collab),
                             float(dt), int(max_neighbors), _p(rc), 0 if rc is None else rc.shape[0], _p(out), _p(nv),
                             _p(lines), _p(nl))
    return {"action": out, "new_vel": nv, "lines": lines[:nl[1]].copy(), "n_obst_lines": int(nl[0])}


# ---- information-gain primitives (one world) --------------------------------------------------------
FOV60 = 60.0 * np.pi / 180


def edt(occupancy):
    m = np.ascontiguousarray(occupancy, dtype=np.uint8)
    edf = np.zeros((300, 300))
    d2 = np.zeros((300, 300), dtype=np.uint32)
    lib().cao_edt(_p(m), _p(edf), _p(d2))
    return edf, d2


def check_visibility(edf, a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return bool(lib().cao_ig_check_visibility(_p(edf), _p(a), _p(b)))


def visible_cells(edf, pose, fov=FOV60, rng=5.0):
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    mask = np.zeros(60, dtype=np.uint64)
    lib().cao_ig_visible(_p(edf), _p(pose), fov, rng, _p(mask))
    return mask


def update_belief(belief, edf, poses, dets, ndet, fov=FOV60, rng=5.0):
    poses = np.ascontiguousarray(poses, dtype=np.float64)
    dets = np.ascontiguousarray(dets, dtype=np.float64)
    ndet = np.ascontiguousarray(ndet, dtype=np.int32)
    obs = np.zeros(60, dtype=np.uint64)
    lib().cao_ig_update(_p(belief), _p(edf), poses.shape[0], _p(poses), _p(ndet), _p(dets), dets.shape[1], fov, rng,
                        _p(obs))
    return obs


def mi_reward(belief, mask):
    mask = np.ascontiguousarray(mask, dtype=np.uint64)
    return lib().cao_ig_reward(_p(belief), _p(mask))


def next_pose(edf, pose, action, xdt=5, dt=0.1, radius=0.5):
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    action = np.ascontiguousarray(action, dtype=np.float64)
    out = np.zeros(3)
    ok = lib().cao_ig_next_pose(_p(edf), _p(pose), _p(action), xdt, dt, radius, _p(out))
    return (out if ok else None)


def rollout(belief, edf, pose0, observed0, exclude, n_steps, seed, q, sim, xdt=5, dt=0.1, radius=0.5, fov=FOV60,
            rng=5.0, want_observed=False):
    pose0 = np.ascontiguousarray(pose0, dtype=np.float64)
    observed0 = np.ascontiguousarray(observed0, dtype=np.uint64)
    exclude = np.ascontiguousarray(exclude, dtype=np.uint64)
    acts = np.zeros(max(n_steps, 1), dtype=np.uint8)
    pose = np.zeros(3)
    obs = np.zeros(60, dtype=np.uint64)
    r = lib().cao_ig_rollout(_p(belief), _p(edf), _p(pose0), _p(observed0), _p(exclude), n_steps, xdt, dt, radius,
                             fov, rng, seed, q, sim, _p(acts), _p(pose), _p(obs))
    if want_observed:
        return r, acts[:n_steps], pose, obs
    return r, acts[:n_steps], pose


GEN_DEFAULTS = dict(n_min=None, n_max=None, ego_policy=5, ego_dynamics=0, policy_a=5, policy_b=1, other_dynamics=0,
                    max_tries=100000, p_b=0.5, side=7.5, min_travel=4.0, min_sep=1.5, radius=0.5, pref_speed=1.0, coop=0.5)


def generate_scenarios(S, M, seed, **kw):
    """CPU twin of cagym_generate_scenarios (train_agents_random_positions, test_cases.py:1362-1463).
    Returns (agents6[S,M,6], policy[S,M], dynamics[S,M], n_agents[S], coop[S,M], n_failed)."""
    P = dict(GEN_DEFAULTS, **kw)
    n_min = M if P["n_min"] is None else P["n_min"]
    n_max = M if P["n_max"] is None else P["n_max"]
    a6 = np.zeros((S, M, 6))
    pol = np.zeros((S, M), dtype=np.int32)
    dyn = np.zeros((S, M), dtype=np.int32)
    na = np.zeros(S, dtype=np.int32)
    cp = np.zeros((S, M))
    nf = lib().cao_generate_scenarios(seed, S, M, n_min, n_max, P["ego_policy"], P["ego_dynamics"], P["policy_a"],
                                      P["policy_b"], P["other_dynamics"], P["max_tries"], P["p_b"], P["side"],
                                      P["min_travel"], P["min_sep"], P["radius"], P["pref_speed"], P["coop"], _p(a6),
                                      _p(pol), _p(dyn), _p(na), _p(cp))
    return a6, pol, dyn, na, cp, nf


def occupancy_grid(static_map, px, py, heading):
    """OccupancyGridSensor.sense (sensors/OccupancyGridSensor.py:70-98) for one agent: static_map [300,300] bool
    (rasterize()), returns the [60,60] bool local grid.  PARITY UNPINNED (cv2 absent): see cagym_oracle_grid.c."""
    m = np.ascontiguousarray(np.asarray(static_map) != 0, dtype=np.uint8)
    out = np.zeros((60, 60), dtype=np.uint8)
    lib().cao_occupancy_grid(_p(m), float(px), float(py), float(heading), _p(out))
    return out.astype(bool)
