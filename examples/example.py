#!/usr/bin/env python3
"""The reference's minimum working example (experiments/src/example.py:22-54) on cagym, line for line:
2 agents, one driven externally, one internal; env.set_agents / reset / step / game_over exactly as there."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cagym  # noqa: E402
from cagym.env import CollisionAvoidanceEnv, Config, get_testcase_two_agents  # noqa: E402

Config.DT = 0.1

env = CollisionAvoidanceEnv()                      # gym.make("CollisionAvoidance-v0")
agents = get_testcase_two_agents()                 # tc.get_testcase_two_agents()
env.set_agents(agents)
obs = env.reset()                                  # agents' initial observations

num_steps = 100
for i in range(num_steps):
    actions = {}
    actions[0] = np.array([1., 0.5])               # external agent 0; internal agents query their own policy
    obs, rewards, game_over, which_agents_done = env.step(actions)
    if game_over:
        print("All agents finished!")
        break
env.reset()
print("Experiment over. steps:", i + 1, "agent 0 at", env.prev_episode_agents[0].pos_global_frame)
