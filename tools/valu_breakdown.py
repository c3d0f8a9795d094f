#!/usr/bin/env python3
"""Workload for the VALU instruction breakdown of the headline kernel: 4096 worlds x 10 agents, 128-step launches, with the
policy (rvo | noncoop) and the observation outputs (obs | noobs) switched from the command line.  Run under
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python3 tools/valu_breakdown.py rvo obs
(tools/valu_breakdown.sh does the four combinations and prints wave-VALU instructions per workgroup-step):
  rows (OAS + scalar observations) = [rvo obs] - [rvo noobs];   ORCA (half-planes, ranking, linear programs, orca_post) =
  [rvo noobs] - [noncoop noobs];   the rest (S1 dynamics, pair distances, S2, ego frame) = [noncoop noobs]."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
pol = scen.POLICY_RVO if sys.argv[1] == "rvo" else scen.POLICY_NONCOOP
obs = sys.argv[2] == "obs"
N, M, T = int(os.environ.get("LC_WORLDS", 4096)), int(os.environ.get("LC_AGENTS", 10)), 128
env = B(N, M, n_scenarios=8 * N, game_over_mode="all")
env.set_scenarios(scen.random_worlds_fast(8 * N, M, seed=1234), pol, scen.DYN_UNICYCLE, coop=np.full((8 * N, M), 0.5))
env.reset()
traj = env.alloc_rollout(T, obs=obs)
for _ in range(12):
    env.rollout(T, out=traj)
torch.cuda.synchronize()
