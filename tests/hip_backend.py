"""Adapter giving the HIP env the backend interface of tests/golden_util.replay."""
import importlib

import numpy as np

cagym = importlib.import_module("gym-exploration-2d_amd")

HIP_FLOAT_KEYS = ["pos", "vel", "heading", "speed", "delta_heading", "dist_to_goal", "heading_ego", "rel_goal",
                  "time_remaining", "t", "reward"]


class HipBackend(object):
    def __init__(self, N, M, max_obstacles=0, game_over_mode=0, laserscan=False, collide_with_static=False,
                 n_scenarios=None, rvo_max_neighbors=0):
        from importlib import import_module
        B = import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
        self.env = B(N, M, n_scenarios=n_scenarios or N, max_obstacles=max_obstacles, game_over_mode=game_over_mode,
                     collide_with_static=collide_with_static, laserscan=laserscan, rvo_max_neighbors=rvo_max_neighbors)
        self.N, self.M = N, M

    def set_scenario(self, agents6, policy_id, dynamics_id, heading0=None, n_agents=None, coop=None,
                     obstacles=None, n_obst=None):
        self.env.set_scenarios(agents6, policy_id, dynamics_id, heading0=heading0, n_agents=n_agents, coop=coop,
                               obstacles=obstacles, n_obst=n_obst)

    def reset(self, world_mask=None):
        self.env.reset(world_mask)

    def step(self, ext=None):
        self.env.step(None if ext is None else np.asarray(ext, dtype=np.float32))

    def f(self, name):
        return self.env.f(name)

    def u(self, name):
        return self.env.u(name)

    def i(self, name):
        return self.env.i(name)
