"""Diagnostic: where the GA3C-CADRL forward kernel's time goes (never the shipped library).  Builds libcagym_hip_gastamps.so
(-DCAGYM_STAMPS -DGA_STAMPS: thread 0 of every workgroup adds s_memtime ticks per phase) and runs cagym_ga3c_act on cfg4's
composition (8192 evaluations = one workgroup per CU) and on 81 920 evaluations."""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch
# extra compiler flags select a variant (its own library: build it in the development container, it travels), e.g. -DGA_DIAG_NOB
TAG = "gastamps" + "".join("_" + a[2:].lower().replace("=", "") for a in sys.argv[1:])
os.environ["CAGYM_LIB"] = b.build_variant(TAG, ["-DCAGYM_STAMPS", "-DGA_STAMPS"] + sys.argv[1:])
print("library", TAG)
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
NAMES = ["prologue (state rows -> operand fragments, LSTM weights)", "LSTM steps", "layer1 68 -> 256", "layer2 256 -> 256", "layer3 256 -> 256", "logits, softmax, action",
         "cagym_ga3c_act only: agent list + state rows in LDS (before the prologue)"]
for name, N, M in (("agent0_8192", 8192, 10), ("all_agents_81920", 8192, 10)):
    pol = np.full((N, M), scen.POLICY_GA3C if "all" in name else scen.POLICY_RVO, dtype=np.int32)
    pol[:, 0] = scen.POLICY_GA3C
    env = B(N, M, game_over_mode="agent0")
    env.set_scenarios(scen.random_worlds_fast(N, M, seed=5), pol, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5))
    env.reset()
    p = GA3C(env)
    ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
    for _ in range(3):
        p.act(ext)
        env.step(ext)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    env.L.cagym_debug_stamps(out, 1)
    R = 20
    for _ in range(R):
        p.act(ext)
    torch.cuda.synchronize()
    env.L.cagym_debug_stamps(out, 0)
    wg = max(1, out[15])
    tot = sum(out[:7])
    print("%s: %d workgroups per launch, s_memtime ticks (shader clock, ~2.2 GHz) per workgroup: total %.1f" % (name, wg // R, tot / wg))
    for i, n in enumerate(NAMES):
        print("  %-40s %9.1f  %5.1f %%" % (n, out[i] / wg, 100.0 * out[i] / max(1, tot)))
    env.close()
