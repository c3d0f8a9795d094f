#!/usr/bin/env python3
"""Who ends cfg4's episodes, and how?  (bench.py --config cfg4 reports frac_collision 0.56 over the agents that carry an outcome
flag when their world's episode ends.)  Runs the cfg4 composition (agent 0 GA3C-CADRL, 9 RVO among 2-10 rectangles, game over =
agent 0 done, auto-reset) and tallies, per finished episode, agent 0's outcome and the other agents' flags, telling wall
collisions (reward -0.25/13) from agent collisions (-10/13)."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
N, M, K, T = 2048, 10, 10, 700
a6, ob, nob, _ = scen.obstacle_worlds(2 * N, M, K, seed=1234)
pol = np.full((2 * N, M), scen.POLICY_RVO, dtype=np.int32)
pol[:, 0] = scen.POLICY_GA3C
env = B(N, M, n_scenarios=2 * N, max_obstacles=K, laserscan=True, game_over_mode="agent0")
env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((2 * N, M), 0.5), obstacles=ob, n_obst=nob)
env.reset()
ga3c = GA3C(env)
ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
AT_GOAL, COLL, TOUT = 1, 4, 16  # CAGYM_FLAG_* (include/cagym.h)
lib = importlib.import_module("gym-exploration-2d_amd._lib")
AT_GOAL, COLL, TOUT = lib.FLAG_AT_GOAL, lib.FLAG_IN_COLLISION, lib.FLAG_RAN_OUT_OF_TIME
tally = {k: 0 for k in ("episodes", "a0_goal", "a0_wall", "a0_agent", "a0_timeout", "others_goal", "others_wall", "others_agent", "others_timeout", "others_unflagged")}
prev = torch.zeros((N, M), dtype=torch.uint8, device=env.device)
kind = torch.zeros((N, M), dtype=torch.int8, device=env.device)  # 1 wall, 2 agent: how a slot's collision happened
for t in range(T):
    ga3c.act(ext)
    env.step(ext, auto_reset=True)
    fl, rw, go = env.flags, env.reward, env.game_over.bool()
    newc = ((fl & COLL) != 0) & ((prev & COLL) == 0)
    kind = torch.where(newc, torch.where(rw > -0.1, torch.ones_like(kind), 2 * torch.ones_like(kind)), kind)
    if go.any():
        f, k = fl[go], kind[go]
        tally["episodes"] += int(go.sum())
        tally["a0_goal"] += int(((f[:, 0] & AT_GOAL) != 0).sum())
        tally["a0_timeout"] += int((((f[:, 0] & TOUT) != 0) & ((f[:, 0] & (AT_GOAL | COLL)) == 0)).sum())
        tally["a0_wall"] += int((((f[:, 0] & COLL) != 0) & (k[:, 0] == 1)).sum())
        tally["a0_agent"] += int((((f[:, 0] & COLL) != 0) & (k[:, 0] == 2)).sum())
        o, ko = f[:, 1:], k[:, 1:]
        tally["others_goal"] += int(((o & AT_GOAL) != 0).sum())
        tally["others_timeout"] += int((((o & TOUT) != 0) & ((o & (AT_GOAL | COLL)) == 0)).sum())
        tally["others_wall"] += int((((o & COLL) != 0) & (ko == 1)).sum())
        tally["others_agent"] += int((((o & COLL) != 0) & (ko == 2)).sum())
        tally["others_unflagged"] += int(((o & (AT_GOAL | COLL | TOUT)) == 0).sum())
    prev = torch.where(go[:, None], torch.zeros_like(fl), fl)
    kind = torch.where(go[:, None], torch.zeros_like(kind), kind)
e = max(1, tally["episodes"])
print("cfg4 composition, %d worlds, %d steps: %d finished episodes" % (N, T, tally["episodes"]))
print("agent 0 (GA3C-CADRL, sees agents only - no obstacle input): goal %.3f  wall %.3f  agent collision %.3f  time-out %.3f"
      % (tally["a0_goal"] / e, tally["a0_wall"] / e, tally["a0_agent"] / e, tally["a0_timeout"] / e))
print("the 9 RVO agents at that moment, per episode: at goal %.2f  wall %.2f  agent collision %.2f  time-out %.2f  still under way %.2f"
      % tuple(tally[k] / e for k in ("others_goal", "others_wall", "others_agent", "others_timeout", "others_unflagged")))
flagged = sum(tally[k] for k in ("a0_goal", "a0_wall", "a0_agent", "a0_timeout", "others_goal", "others_wall", "others_agent", "others_timeout"))
print("share of collisions among the flagged agents (the bench line's frac_collision): %.3f"
      % ((tally["a0_wall"] + tally["a0_agent"] + tally["others_wall"] + tally["others_agent"]) / max(1, flagged)))
