#!/usr/bin/env python3
"""Diagnostic: where does the cfg4 step (8192 x 10, per-step launches) spend its time?  Times cagym_step_autoreset for
variants of the workload (rectangles / RVO / LaserScan switched on one at a time)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
N, M, K = 8192, 10, 10
a6, ob, nob, _ = scen.obstacle_worlds(2 * N, M, K, seed=3)
free = scen.random_worlds_fast(2 * N, M, seed=3)


def run(name, policy, obstacles, laser, agents6):
    env = B(N, M, n_scenarios=2 * N, max_obstacles=K if obstacles else 0, laserscan=laser, game_over_mode="agent0")
    pol = np.full((2 * N, M), policy, dtype=np.int32)
    pol[:, 0] = scen.POLICY_EXTERNAL
    if obstacles:
        env.set_scenarios(agents6, pol, scen.DYN_UNICYCLE, coop=np.full((2 * N, M), 0.5), obstacles=ob, n_obst=nob)
    else:
        env.set_scenarios(agents6, pol, scen.DYN_UNICYCLE, coop=np.full((2 * N, M), 0.5))
    env.reset()
    ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
    ext[:, 0, 0] = 1.0
    for _ in range(30):
        env.step(ext, auto_reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 100
    for _ in range(R):
        env.step(ext, auto_reset=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print("%-70s %8.3f ms/step   %s" % (name, dt * 1e3, env.kernel_name(rollout=False, auto_reset=True)))
    env.close()


run("free space, 9 RVO, no laser", scen.POLICY_RVO, False, False, free)
run("free space, 9 NonCooperative, no laser", scen.POLICY_NONCOOP, False, False, free)
run("rectangles (ring scenario), 9 NonCooperative, no laser (wall test only)", scen.POLICY_NONCOOP, True, False, a6)
run("rectangles, 9 NonCooperative, LaserScan", scen.POLICY_NONCOOP, True, True, a6)
run("rectangles, 9 RVO (obstacle half-planes), no laser", scen.POLICY_RVO, True, False, a6)
run("rectangles, 9 RVO, LaserScan (= cfg4 env part)", scen.POLICY_RVO, True, True, a6)
