#!/usr/bin/env python3
"""A stable-baselines style loop (experiments/src/env_utils.py:29-62) on the vectorised env: 4096 worlds, the learning
agent (agent 0 of every world) driven by a random policy on the GPU, the other agents by ORCA inside the step kernel;
observations are the flat MultiagentFlattenDictWrapper layout, finished worlds restart inside the step launch."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
vec = importlib.import_module("gym-exploration-2d_amd.vecenv")

N, M = 4096, 10
env = B(N, M, n_scenarios=8 * N, game_over_mode="agent0")
env.generate_scenarios(seed=1, ego_policy=scen.POLICY_LEARNING, other_policies=(scen.POLICY_RVO, scen.POLICY_NONCOOP), p_b=0.2)
KEYS = ['dist_to_goal', 'rel_goal', 'radius', 'heading_ego_frame', 'pref_speed', 'other_agents_states']  # config.py:97
venv = vec.CagymVecEnv(env, KEYS)
obs = venv.reset()
print("flat observation:", tuple(obs.shape), obs.dtype, obs.device)
ret = torch.zeros(N, device=env.device)
t0 = time.perf_counter()
steps = 500
for _ in range(steps):
    actions = torch.zeros((N, M, 2), device=env.device)
    actions[:, 0] = torch.rand((N, 2), device=env.device)    # network outputs in [0, 1]^2 (LearningPolicy.py:11-16)
    venv.step_async(actions)
    obs, rew, done, infos = venv.step_wait()
    ret += rew
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d steps of %d worlds in %.2f s = %.1f M env-steps/s; mean reward per step %.5f" % (steps, N, dt, steps * N / dt / 1e6, float(ret.mean()) / steps))
