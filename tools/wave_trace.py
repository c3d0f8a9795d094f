#!/usr/bin/env python3
"""Diagnostic (never the shipped library): per-wave s_memtime stamps of one workgroup of the generation-3 rollout kernel
(-DCAGYM_WAVETRACE).  Prints, per phase, when each wave reaches the marked points relative to the step's start: the wave
that arrives last at a barrier is the step's critical chain."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch
if "--child" not in sys.argv:
    # one monolithic diagnostic unit (build.build_variant caches it: build it in the development container, the GPU box reuses it)
    LIB = b.build_variant("wavetrace", ["-DCAGYM_WAVETRACE"] + os.environ.get("WT_DEFS", "").split())
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, CAGYM_LIB=LIB)))
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
N, M, T = int(os.environ.get("LC_WORLDS", 4096)), 10, 24
if os.environ.get("LC_OBST"):  # cfg4's env part as a roll-out: RVO agents among 2-10 rectangles (OBST instantiation)
    a6, ob, nob, _ = scen.obstacle_worlds(2 * N, M, 10, seed=3)
    env = B(N, M, n_scenarios=2 * N, max_obstacles=10, game_over_mode="all", laserscan=bool(os.environ.get("LC_LASER")))
    env.set_scenarios(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((2 * N, M), 0.5), obstacles=ob, n_obst=nob)
else:
    env = B(N, M, n_scenarios=8 * N, game_over_mode="all")
    env.set_scenarios(scen.random_worlds_fast(8 * N, M, seed=1234), scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((8 * N, M), 0.5))
env.reset()
traj = env.alloc_rollout(64)
for _ in range(6):
    env.rollout(64, out=traj)
P, W = 16, 8
acc = []
for rep in range(12):
    env.rollout(T, out=traj)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (T * P * W))()
    env.L.cagym_debug_wavetrace(buf)
    X = np.frombuffer(buf, dtype=np.uint64).reshape(T, P, W).astype(np.float64)[:, :, :4]
    t0 = X[:, 0, :].min(axis=1)[:, None, None]
    acc.append(X[1:] - t0[1:])  # step 0 has no rows to write
    env.rollout(40, out=traj)
A = np.concatenate(acc)  # [steps, points, waves] ticks since the first wave entered the step
names = ["step top", "busy list built", "own LP groups done", "wave 0: all LP waves done", "S1 done (w0) / rows done (others)", "after barrier X",
         "after publish + barrier Y", "last wave: ego frame + LP inputs done", "pair distances done", "after barrier A",
         "S2 done (w0) / half-planes done (others)", "after barrier B", "step end", "LP: lines ranked, own lines loaded", "LP: first program (linearProgram2) done", "S1: action chosen (orca_post)"]
print("%d worlds, workgroup 7, median over %d steps; ticks since the step's first wave started (0 = point not reached by that wave)" % (N, A.shape[0]))
print("%-46s %9s %9s %9s %9s" % ("point", "wave 0", "wave 1", "wave 2", "wave 3"))
for k, n in enumerate(names):
    v = A[:, k, :]
    med = [np.median(v[:, w][v[:, w] > 0]) if (v[:, w] > 0).any() else 0 for w in range(4)]
    print("%-46s %9.0f %9.0f %9.0f %9.0f" % (n, *med))
