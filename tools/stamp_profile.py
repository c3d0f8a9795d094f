#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the phase-split step kernel (never the shipped library).
Builds libcagym_hip_stamps.so with -DCAGYM_STAMPS, runs a rollout, prints cycles per step per phase."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch
lib = b.build_variant("stamps", ["-DCAGYM_STAMPS"])
os.environ["CAGYM_LIB"] = lib
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
pol = scen.POLICY_RVO if "noncoop" not in sys.argv else scen.POLICY_NONCOOP
N, M, T = int(os.environ.get("LC_WORLDS", 4096)), 10, 64
if "obst" in sys.argv:  # cfg4's env part as a roll-out: RVO agents among 2-10 rectangles (OBST instantiation), no laser
    a6, ob, nob, _ = scen.obstacle_worlds(2 * N, M, 10, seed=3)
    env = B(N, M, n_scenarios=2 * N, max_obstacles=10, game_over_mode="all")
    env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((2 * N, M), 0.5), obstacles=ob, n_obst=nob)
else:
    env = B(N, M, n_scenarios=8 * N, game_over_mode="all")
    env.set_scenarios(scen.random_worlds_fast(8 * N, M), pol, scen.DYN_UNICYCLE, coop=np.full((8 * N, M), 0.5))
env.reset()
traj = env.alloc_rollout(T)
for _ in range(4):
    env.rollout(T, out=traj)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
env.L.cagym_debug_stamps(out, 1)
R = 8
for _ in range(R):
    env.rollout(T, out=traj)
torch.cuda.synchronize()
env.L.cagym_debug_stamps(out, 0)
gen = os.environ.get("CAGYM_KERNEL", "v3")
if gen in ("v2", "2"):
    names = ["S0 barrier", "P1 half-planes per unordered pair+bar", "(unused)", "S1 action maps + dynamics+bar", "P2 pair distances+bar",
             "S2 reward/done/reset+bar", "P3 OAS rows", "ego observation store", "LP: wait for the other groups + barrier", "(unused)",
             "LP: list + ranking (group 0)", "LP: linearProgram2/3 of group 0's ego"]
else:
    names = ["obstacle lines 1: (ego, rectangle) candidate tests (part of A)", "C: busy list + linear programs of wave 0", "D: wait for the other LP waves", "D: S1 action maps + dynamics (registers)",
             "barrier (row workers) + publish moved state + barrier", "A: pair distances/keys/dsq + barrier", "B: S2 reward/done/reset (+ tail pairs) + barrier (half-planes on waves 1..)",
             "reset rebuild (rare)", "C: busy list", "obstacle lines 2: per-ego sort + work list (part of A)", "obstacle lines 3: half-plane per (ego, candidate) (part of A)", "(unused)"]
if gen not in ("v2", "2"):
    print("wave 0 of workgroup 0, per step: linearProgram1 calls in linearProgram2 (longest group) %.2f, linearProgram3 outer %.3f, inner %.3f; busy egos of the workgroup %.1f"
          % tuple(out[12 + k] / (R * T) for k in range(4)))
tot = sum(out[:12])
print("s_memtime ticks per step, workgroup 0; total %.1f per step (%s kernels, %d worlds)" % (tot / (R * T), gen, N))
for i, n in enumerate(names[:12]):
    if out[i]:
        print("  %-80s %10.1f  %5.1f %%" % (n, out[i] / (R * T), 100.0 * out[i] / tot))
