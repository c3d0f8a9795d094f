#!/usr/bin/env python3
"""Kernel timeline of cfg4's overlapped split step, from rocprofv3's kernel trace.
  step 1 (GPU box):  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/ovtrace -o run -- python3 tools/cfg4_overlap_trace.py run [variant]
  step 2 (anywhere): python tools/cfg4_overlap_trace.py parse gpurun_out/r4/ovtrace
`run` issues 40 steps of the chosen variant (policy_first | pre_first | serial | fused); `parse` prints, for the last steps, when each
kernel started and ended relative to the step's first kernel (us)."""
import csv, glob, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "parse":
    files = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?")))
    rows.sort()
    names = ("k_ga3c_select", "k_ga3c_state", "k_ga3c_forward", "k_step_pre3", "k_step_post3", "k_step3")
    rows = [r for r in rows if any(n in r[2] for n in names)]
    # a step starts with k_ga3c_select or k_step_pre3 after a k_step_post3 / k_step3
    steps, cur = [], []
    for r in rows:
        if cur and ("k_step_post3" in cur[-1][2] or "void k_step3" in cur[-1][2] or cur[-1][2].startswith("k_step3")):
            steps.append(cur)
            cur = []
        cur.append(r)
    for st in steps[-4:]:
        t0 = min(r[0] for r in st)
        print("step: span %.1f us" % ((max(r[1] for r in st) - t0) / 1e3))
        for s, e, n, q in st:
            print("   %-60s queue %s  start %7.1f  end %7.1f  (%.1f us)" % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    if len(steps) > 8:
        import statistics
        sp = [(max(r[1] for r in st) - min(r[0] for r in st)) / 1e3 for st in steps[5:]]
        gaps = [(steps[i + 1][0][0] - max(r[1] for r in steps[i])) / 1e3 for i in range(5, len(steps) - 1)]
        print("median step span %.1f us over %d steps; median gap to the next step's first kernel %.1f us" % (statistics.median(sp), len(sp), statistics.median(gaps)))
    sys.exit(0)

import numpy as np
import torch
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
variant = sys.argv[2] if len(sys.argv) > 2 else "policy_first"
N, M, K = 8192, 10, 10
S = 2 * N
a6, ob, nob, _ = scen.obstacle_worlds(S, M, K, seed=1234)
pol = np.full((S, M), scen.POLICY_RVO, dtype=np.int32)
pol[:, 0] = scen.POLICY_GA3C
env = B(N, M, n_scenarios=S, max_obstacles=K, laserscan=True, game_over_mode="agent0")
env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((S, M), 0.5), obstacles=ob, n_obst=nob)
env.reset()
ga3c = GA3C(env)
ext = torch.zeros((N, M, 2), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for _ in range(40):
    if variant == "fused":
        ga3c.act(ext)
        env.step(ext, auto_reset=True)
    elif variant == "serial":
        env.step_begin()
        ga3c.act(ext)
        env.step_finish(ext, auto_reset=True)
    else:
        side.wait_stream(main)
        if variant == "pre_first":
            env.step_begin(stream=side)
            ga3c.act(ext)
        else:
            ga3c.act(ext)
            env.step_begin(stream=side)
        main.wait_stream(side)
        env.step_finish(ext, auto_reset=True)
torch.cuda.synchronize()
