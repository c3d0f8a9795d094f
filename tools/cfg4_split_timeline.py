#!/usr/bin/env python3
"""Diagnostic (never the shipped library): per-workgroup sub-phase timeline of the split step's two launches on cfg4
(k_step_pre3 / k_step_post3, -DCAGYM_WGTRACE build: thread 0 of every workgroup stamps the 100 MHz s_memrealtime clock).
Prints each sub-phase's median / p90 over the workgroups, the workgroups' own durations and their entry times (= the rounds of
a launch that is not co-resident)."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch
if "--child" not in sys.argv:
    LIB = b.build_variant("wgtrace", ["-DCAGYM_WGTRACE"])
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, CAGYM_LIB=LIB)))
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
N, M, K = int(os.environ.get("LC_WORLDS", 8192)), 10, 10
S = 2 * N
a6, ob, nob, _ = scen.obstacle_worlds(S, M, K, seed=1234)
pol = np.full((S, M), scen.POLICY_RVO, dtype=np.int32)
pol[:, 0] = scen.POLICY_GA3C
env = B(N, M, n_scenarios=S, max_obstacles=K, laserscan=True, game_over_mode="agent0")
env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((S, M), 0.5), obstacles=ob, n_obst=nob)
env.reset()
ga3c = GA3C(env)
ext = torch.zeros((N, M, 2), dtype=torch.float32, device="cuda")
for _ in range(60):
    ga3c.act(ext)
    env.step(ext, auto_reset=True)
torch.cuda.synchronize()
W = 48
wpw = 4 if "10, 4" in env.kernel_name(rollout=False, auto_reset=True) else 5
n_wg = min((N + wpw - 1) // wpw, 4096)
buf = (ctypes.c_ulonglong * (W * n_wg))()
PRE = [(0, "entry"), (20, "ten state fields HBM -> LDS, barrier"), (21, "rectangles staged"), (33, "LP inputs; obstacle lines 1: edge tests -> candidate lists"),
       (34, "obstacle lines 2: work list (one wave)"), (35, "obstacle lines 3: rank + half-plane per (ego, candidate)"), (36, "obstacle lines 4: coverage bits"),
       (22, "obstacle lines 5: per-ego walk (one wave)"), (23, "neighbour keys of the pairs, barrier"), (24, "agent half-planes + ranking, barrier"),
       (25, "busy list + linear programs, barrier"), (38, "8 B per agent -> HBM, exit")]
POST = [(0, "entry"), (20, "agent records HBM -> LDS, barrier"), (21, "rectangles staged"), (1, "prologue ends"), (24, "C: nothing (no RVO work left)"),
        (25, "D: S1 on wave 0 (actions of every policy), barrier"), (26, "publish, barrier"), (40, "A: wall prep (wave 0), pair distances, barrier"),
        (41, "A: wall rows"), (27, "A: barrier"), (28, "B: S2 (+ auto-reset) beside the LaserScan, barrier"), (2, "reset rebuild (rare)"),
        (38, "epilogue: OAS rows (+ scans of restarted worlds) + state -> HBM")]


def trace(launch, pts, title):
    acc = {k: [] for k, _ in pts[1:]}
    spans, own, entries = [], [], []
    for rep in range(20):
        ga3c.act(ext)
        if launch == "pre":
            torch.cuda.synchronize()
            env.step_begin()
            torch.cuda.synchronize()
            env.L.cagym_debug_wgtrace(buf, n_wg)
            T = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, W).astype(np.int64).copy()
            env.step_finish(ext, auto_reset=True)
        else:
            env.step_begin()
            torch.cuda.synchronize()
            env.step_finish(ext, auto_reset=True)
            torch.cuda.synchronize()
            env.L.cagym_debug_wgtrace(buf, n_wg)
            T = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, W).astype(np.int64).copy()
        t0 = T[:, 0].min()
        prev = T[:, 0]
        for k, _ in pts[1:]:
            acc[k].append((T[:, k] - prev) * 0.01)
            prev = T[:, k]
        spans.append((T[:, 38].max() - t0) * 0.01)
        own.append((T[:, 38] - T[:, 0]) * 0.01)
        entries.append(np.sort((T[:, 0] - t0) * 0.01))
    own = np.concatenate(own)
    print("%s, %d worlds = %d workgroups; 20 launches; kernel span (first entry -> last exit) median %.1f us" % (title, N, n_wg, np.median(spans)))
    print("a workgroup's own duration: median %.1f  p10 %.1f  p90 %.1f  max %.1f us" % (np.median(own), np.percentile(own, 10), np.percentile(own, 90), own.max()))
    e = np.median(np.stack(entries), axis=0)
    print("workgroup entry times (sorted, median over launches): " + "  ".join("%d%%: %.1f" % (q, e[min(n_wg - 1, int(q / 100 * n_wg))]) for q in (0, 10, 25, 40, 49, 51, 60, 75, 90, 99)) + " us")
    print("%-72s %8s %8s %8s" % ("sub-phase (thread 0's stamps)", "median", "p90", "share"))
    tot = sum(np.median(np.concatenate(acc[k])) for k, _ in pts[1:])
    for k, n in pts[1:]:
        v = np.concatenate(acc[k])
        print("%-72s %8.2f %8.2f %7.1f%%" % (n, np.median(v), np.percentile(v, 90), 100 * np.median(v) / tot))
    print()


trace("pre", PRE, "k_step_pre3<256, 10, %d, true>" % wpw)
trace("post", POST, "k_step_post3<256, 10, %d, true, true>" % wpw)
