#!/usr/bin/env python3
"""Diagnostic (never the shipped library): which phase makes the slowest workgroup of a 20-step launch slow?
Builds with -DCAGYM_WGTRACE -DCAGYM_WAVETRACE, finds the workgroup that finished last in a 20-step launch (crowds persist from
launch to launch), selects it for the per-wave trace and prints its phase times next to those of a median workgroup."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch
if "--child" not in sys.argv:
    LIB = b.build_variant("slowwg", ["-DCAGYM_WAVETRACE", "-DCAGYM_WGTRACE"])  # one monolithic diagnostic unit, cached (build.py)
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, CAGYM_LIB=LIB)))
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
N, M, T = 4096, 10, 20
env = B(N, M, n_scenarios=8 * N, game_over_mode="all")
env.set_scenarios(scen.random_worlds_fast(8 * N, M, seed=1234), scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((8 * N, M), 0.5))
env.reset()
traj = env.alloc_rollout(64)
for _ in range(6):
    env.rollout(64, out=traj)
n_wg, W, P = 1024, 48, 16
names = ["step top", "busy list built", "own LP groups done", "wave 0: all LP waves done", "S1 done (w0) / rows done (others)", "after barrier X",
         "after publish + barrier Y", "last wave: ego frame + LP inputs done", "pair distances done", "after barrier A",
         "S2 done (w0) / half-planes done (others)", "after barrier B", "step end", "LP: lines loaded, start point", "LP: linearProgram2 done", "S1: action chosen (orca_post)"]


def wg_times():
    buf = (ctypes.c_ulonglong * (W * n_wg))()
    env.L.cagym_debug_wgtrace(buf, n_wg)
    Tm = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, W).astype(np.int64)
    return (Tm[:, 38] - Tm[:, 0]) * 0.01, np.diff(Tm[:, 1:2 + T], axis=1) * 0.01


def trace(wg):
    env.L.cagym_debug_wavetrace_select(int(wg))
    env.rollout(T, out=traj)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (24 * P * 8))()
    env.L.cagym_debug_wavetrace(buf)
    X = np.frombuffer(buf, dtype=np.uint64).reshape(24, P, 8).astype(np.float64)[:T, :, :4]
    t0 = X[:, 0, :].min(axis=1)[:, None, None]
    dur, steps = wg_times()
    return X[1:] - t0[1:], dur[wg], steps[wg]


env.rollout(T, out=traj)
torch.cuda.synchronize()
dur, _ = wg_times()
order = np.argsort(dur)
for label, wg in (("slowest", order[-1]), ("2nd slowest", order[-2]), ("median", order[n_wg // 2])):
    A, d, st = trace(wg)
    print("%s workgroup %d: %.1f us in the selecting launch, %.1f us in the traced one; its steps (us): %s" % (label, wg, dur[wg], d, " ".join("%.1f" % v for v in st)))
    print("  median over its %d traced steps, ticks since the step's first wave started (0 = not reached by that wave)" % A.shape[0])
    print("  %-46s %9s %9s %9s %9s" % ("point", "wave 0", "wave 1", "wave 2", "wave 3"))
    for k, n in enumerate(names):
        v = A[:, k, :]
        med = [np.median(v[:, w][v[:, w] > 0]) if (v[:, w] > 0).any() else 0 for w in range(4)]
        print("  %-46s %9.0f %9.0f %9.0f %9.0f" % (n, *med))
    v = A[:, 12, 0]
    print("  step length (wave 0, ticks): " + " ".join("%.0f" % x for x in v))
