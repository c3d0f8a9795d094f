#!/usr/bin/env python3
"""experiments/src/dmcts.py (the reference's default experiment: IG_agent_crossing, 3 ig_mcts robots + 2 static targets)
for MANY worlds at once: batched env + belief update + team MI reward + Dec-MCTS trees, all on the GPU.
Every world runs the same scenario with its own random streams; prints the cumulative team reward statistics.

usage: python examples/dmcts_experiment.py [--worlds 256] [--steps 30] [--Ntree 30] [--Ncycles 5] [--cp 1.0]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
igm = importlib.import_module("gym-exploration-2d_amd.ig")
dm = importlib.import_module("gym-exploration-2d_amd.dmcts")

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, default=256)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--Ntree", type=int, default=30)
ap.add_argument("--Ncycles", type=int, default=5)
ap.add_argument("--Nsims", type=int, default=10)
ap.add_argument("--cp", type=float, default=1.0)
args = ap.parse_args()

N, M = args.worlds, 10
OBST = [(2, 2, 10, 10), (-10, 2, -2, 10), (2, -10, 10, -2), (-10, -10, -2, -2)]  # test_cases.py:3219-3222
a6 = np.zeros((M, 6))
a6[:, 4], a6[:, 5], a6[:, 0] = 1.0, 0.1, 1e3 + np.arange(M)
a6[0], a6[1], a6[2] = [-5, 0, 16, 0, 1, .5], [0, 0, 16, 0, 1, .5], [5, 0, 16, 0, 1, .5]   # test_cases.py:3226-3232
a6[3], a6[4] = [6, 12, 0, 0, 1, .2], [-6, -12, 0, 0, 1, .2]                               # static targets
pol = np.zeros(M, dtype=np.int32)
pol[:3] = scen.POLICY_IGMCTS
env = B(N, M, max_obstacles=4, game_over_mode="agent0")
env.set_scenarios(np.tile(a6[None], (N, 1, 1)), np.tile(pol[None], (N, 1)), scen.DYN_FIRSTORDER, heading0=np.zeros((N, M)),
                  n_agents=[5] * N, obstacles=np.tile(np.array(OBST, dtype=np.float64)[None], (N, 1, 1)), n_obst=[4] * N)
env.reset()
ig = igm.InfoGain(env)                                            # detect_fov 60 deg, range 5 m, xdt 5 (dmcts.py:74-78)
planner = dm.DeviceDecMCTSPlanner(ig, 3, radius=0.5, Ntree=args.Ntree, Nsims=args.Nsims, horizon=4, c_p=args.cp,
                                  gamma=0.95, Ncycles=args.Ncycles, seed=0)
world = torch.arange(N, dtype=torch.int32, device=env.device)
targets = torch.tensor(a6[3:5, 0:2], device=env.device)
cum = torch.zeros(N, dtype=torch.float64, device=env.device)
ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
for t in range(args.steps):
    st = env.state()
    poses = torch.stack([st["pos_x"][:, :3], st["pos_y"][:, :3], st["heading"][:, :3]], dim=2)
    # detector emulation (ig_mcts.find_targets_in_obs, quirk Q24): every target within 5 m of a robot is detected
    d = torch.linalg.norm(poses[:, :, None, :2] - targets[None, None], dim=3)       # [N, 3, 2]
    det = targets[None, None].expand(N, 3, 2, 2).contiguous()
    order = torch.argsort((d >= 5.0).to(torch.int8), dim=2, stable=True)           # detected targets first
    det = torch.gather(det, 2, order[..., None].expand(N, 3, 2, 2))
    nd = (d < 5.0).sum(dim=2).to(torch.int32)
    obs = ig.update_belief(poses, det, nd)
    cum += ig.mi_reward(obs, world)                                                # policy.team_reward (dmcts.py:90)
    actions, _ = planner.plan(poses)
    ext[:, :3] = actions.float()
    env.step(ext)
torch.cuda.synchronize()
c = cum.cpu().numpy()
print("worlds %d, steps %d: cumulative team reward mean %.3f, std %.3f, min %.3f, max %.3f" % (N, args.steps, c.mean(), c.std(), c.min(), c.max()))
