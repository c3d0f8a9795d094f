#!/usr/bin/env python3
"""Diagnostic (never the shipped library): where a grow of the device Dec-MCTS planner spends its time.  Builds
libcagym_hip_dmstamps.so (-DCAGYM_STAMPS -DDM_STAMPS: thread 0 of every workgroup adds its s_memtime ticks per phase) and runs
cfg5's planning step (2048 worlds x 3 robots, Ntree 30, Nsims 10, horizon 4, Ncycles 5)."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch
TAG = "dmstamps" + "".join("_" + a[2:].lower().replace("=", "") for a in sys.argv[1:])
os.environ["CAGYM_LIB"] = b.build_variant(TAG, ["-DCAGYM_STAMPS", "-DDM_STAMPS"] + sys.argv[1:])
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
IG = importlib.import_module("gym-exploration-2d_amd.ig").InfoGain
dmm = importlib.import_module("gym-exploration-2d_amd.dmcts")
N, M, K, R = int(os.environ.get("LC_WORLDS", 2048)), 20, 8, 3
S = 2 * N
a6, ob, nob, _ = scen.obstacle_worlds(S, M, K, seed=1234)
pol = np.full((S, M), scen.POLICY_NONCOOP, dtype=np.int32)
pol[:, :3] = scen.POLICY_IGMCTS
pol[:, 3:5] = scen.POLICY_STATIC
dyn = np.full((S, M), scen.DYN_UNICYCLE, dtype=np.int32)
dyn[:, :3] = scen.DYN_FIRSTORDER
env = B(N, M, n_scenarios=S, max_obstacles=K, game_over_mode="all")
env.set_scenarios(a6, pol, dyn, obstacles=ob, n_obst=nob)
env.reset()
ig = IG(env)
planner = dmm.DeviceDecMCTSPlanner(ig, R, radius=0.5, Ntree=30, Nsims=10, horizon=4, c_p=1.0, gamma=0.95, Ncycles=5, seed=1)
st = env.state()
poses = torch.stack([st["pos_x"][:, :R], st["pos_y"][:, :R], st["heading"][:, :R]], dim=2).contiguous()
planner.plan(poses)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
env.L.cagym_debug_stamps(out, 1)
t0 = time.perf_counter()
planner.plan(poses)
torch.cuda.synchronize()
el = time.perf_counter() - t0
env.L.cagym_debug_stamps(out, 0)
NAMES = ["other robots' sampled plans -> excluded cells", "selection (UCT walk)", "masks of the selected node (visibility, block)", "expansion",
         "roll-outs: start + bookkeeping of wave 0's roll-outs", "wait for the slowest wave", "best roll-out + back-propagation", "top-n distribution",
         "roll-outs: motion primitives (next pose, 5 sub-steps)", "roll-outs: visibility queries (cone cells + sphere traces)", "roll-outs: MI reward sums"]
grows = max(1, out[15])
tot = sum(out[:11])
print("planning step %.1f ms (stamped build); %d grows stamped = %d per world; s_memtime ticks of thread 0 per grow: %.0f (= %.1f us at the planning step's duration)" % (1e3 * el, grows, grows // N, tot / grows, 1e6 * el / (grows // N)))
for i, n in enumerate(NAMES):
    print("  %-52s %9.1f ticks  %5.1f %%" % (n, out[i] / grows, 100.0 * out[i] / max(1, tot)))
