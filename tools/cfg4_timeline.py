#!/usr/bin/env python3
"""Diagnostic (never the shipped library): where does one cfg4 env launch (k_step3, OBST instantiation, ONE step per launch) spend
its time?  -DCAGYM_WGTRACE build: thread 0 of every workgroup stamps the 100 MHz s_memrealtime clock at kernel entry, at the
sub-phases of the prologue and of the step, and at exit.  Prints the median / p90 duration of each sub-phase over the
workgroups and how the workgroups' entries spread over the launch (the "rounds" of a launch that is not co-resident)."""
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
name = env.kernel_name(rollout=False, auto_reset=True)
wpw = 4 if "10, 4" in name else 5
n_wg = min((N + wpw - 1) // wpw, 4096)
buf = (ctypes.c_ulonglong * (W * n_wg))()
pts = [(0, "entry"), (20, "agent records HBM -> LDS, barrier"), (21, "rectangles staged"), (33, "LP inputs; obstacle lines 1: (ego, rectangle) edge tests -> candidate lists"), (34, "obstacle lines 2: work list (prefix sums, one wave)"),
       (35, "obstacle lines 3: rank + half-plane per (ego, candidate)"), (36, "obstacle lines 4: coverage bits per (ego, candidate, earlier line)"),
       (22, "obstacle lines 5: per-ego walk, compaction (one wave)"),
       (23, "pair distances, barrier"), (1, "agent half-planes, barrier (prologue ends)"), (24, "C: busy list + linear programs"),
       (25, "D: S1 on wave 0 (no rows: first step), barrier"), (26, "publish, barrier"), (40, "A: wall prep (wave 0), pair distances, barrier"), (41, "A: wall rows (thread 0's wave)"), (27, "A: barrier"),
       (28, "B: S2 (+ auto-reset), barrier"), (2, "reset rebuild (rare)"), (38, "epilogue: OAS rows + laser scans + state -> HBM")]
acc = {k: [] for k, _ in pts[1:]}
spans, own, entries, las, ntodo, busy = [], [], [], [], [], []
for rep in range(20):
    ga3c.act(ext)
    torch.cuda.synchronize()
    env.step(ext, auto_reset=True)
    torch.cuda.synchronize()
    env.L.cagym_debug_wgtrace(buf, n_wg)
    T = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, W).astype(np.int64)
    t0 = T[:, 0].min()
    prev = T[:, 0]
    for k, _ in pts[1:]:
        acc[k].append((T[:, k] - prev) * 0.01)
        prev = T[:, k]
    spans.append((T[:, 38].max() - t0) * 0.01)
    own.append((T[:, 38] - T[:, 0]) * 0.01)
    entries.append(np.sort((T[:, 0] - t0) * 0.01))
    las.append(np.stack([(T[:, 30] - T[:, 29]) * 0.01, (T[:, 31] - T[:, 30]) * 0.01, T[:, 32].astype(np.float64)], 1))
    ntodo.append(T[:, 37].copy())
    busy.append((T[:, 39] >> 8).copy())
own = np.concatenate(own)
print("%s, %d worlds = %d workgroups; 20 launches; kernel span (first entry -> last exit) median %.1f us" % (name, N, n_wg, np.median(spans)))
print("a workgroup's own duration: median %.1f  p10 %.1f  p90 %.1f  max %.1f us" % (np.median(own), np.percentile(own, 10), np.percentile(own, 90), own.max()))
e = np.median(np.stack(entries), axis=0)
print("workgroup entry times (sorted, median over launches): " + "  ".join("%d%%: %.1f" % (q, e[min(n_wg - 1, int(q / 100 * n_wg))]) for q in (0, 10, 25, 37, 38, 50, 62, 75, 76, 90, 99)) + " us")
print("%-66s %8s %8s %8s" % ("sub-phase (thread 0's stamps)", "median", "p90", "share"))
tot = sum(np.median(np.concatenate(acc[k])) for k, _ in pts[1:])
for k, n in pts[1:]:
    v = np.concatenate(acc[k])
    print("%-66s %8.2f %8.2f %7.1f%%" % (n, np.median(v), np.percentile(v, 90), 100 * np.median(v) / tot))
las = np.concatenate(las)
print("LaserScan, wave 1's share: the slab-test passes it claimed (of %d, 64 beams each) median %.2f p90 %.2f us; wait + the sampling rounds it claimed median %.2f p90 %.2f us; "
      "beams listed by the workgroup median %d p90 %d of %d" % ((wpw * M + 3) // 4, np.median(las[:, 0]), np.percentile(las[:, 0], 90), np.median(las[:, 1]), np.percentile(las[:, 1], 90),
                                               np.median(las[:, 2]), np.percentile(las[:, 2], 90), wpw * M * 16))
nt = np.concatenate(ntodo)
print("obstacle lines: (ego, candidate) work items per workgroup median %d p90 %d; x longest candidate list = coverage items" % (np.median(nt & 0xffff), np.percentile(nt & 0xffff, 90)))
print("                longest candidate list median %d p90 %d" % (np.median(nt // 65536), np.percentile(nt // 65536, 90)))
bz = np.concatenate(busy)
print("busy egos (LP groups) per workgroup: median %d p90 %d max %d; workgroups with more than 32 (a second round of LP groups): %.1f %%" % (np.median(bz), np.percentile(bz, 90), bz.max(), 100.0 * (bz > 32).mean()))
