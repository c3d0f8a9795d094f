#!/usr/bin/env python3
"""cfg4's step, piece by piece and together (round 4): cagym_ga3c_act, the split step's two launches (cagym_step_begin /
cagym_step_finish) and the fused launch alone, then the whole step in its variants - fused, split on one stream, split with the
PRE half on a side stream beside the policy (either enqueue order) - each timed over `reps` steps between two synchronisations.
CAGYM_PRE_LDS=<bytes> (read at handle creation) caps the PRE half's workgroups per CU.
usage: python tools/cfg4_overlap.py [reps]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
side = torch.cuda.Stream()


def fused():
    ga3c.act(ext)
    env.step(ext, auto_reset=True)


def split_serial():
    env.step_begin()
    ga3c.act(ext)
    env.step_finish(ext, auto_reset=True)


def split_pre_first():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    env.step_begin(stream=side)
    ga3c.act(ext)
    main.wait_stream(side)
    env.step_finish(ext, auto_reset=True)


def split_policy_first():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ga3c.act(ext)
    env.step_begin(stream=side)
    main.wait_stream(side)
    env.step_finish(ext, auto_reset=True)


def pre_beside_policy_only():  # the overlapped pair without the POST half (state does not advance)
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ga3c.act(ext)
    env.step_begin(stream=side)
    main.wait_stream(side)


def loop(fn, n=reps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for _ in range(60):
    fused()
print("cfg4 %d worlds x %d agents; CAGYM_PRE_LDS=%s; us per call, %d calls between two synchronisations" % (N, M, os.environ.get("CAGYM_PRE_LDS", "-"), reps))
print("pieces alone:   cagym_ga3c_act %.1f   cagym_step_begin %.1f   begin + finish %.1f   cagym_step_autoreset (fused) %.1f" % (
    loop(lambda: ga3c.act(ext)), loop(env.step_begin), loop(lambda: (env.step_begin(), env.step_finish(ext, auto_reset=True))), loop(lambda: env.step(ext, auto_reset=True))))
print("PRE beside the policy (fork / join, no POST): %.1f" % loop(pre_beside_policy_only))
for name, fn in (("fused: ga3c_act, step_autoreset", fused), ("split, one stream: begin, ga3c_act, finish", split_serial),
                 ("split, PRE on a side stream enqueued BEFORE the policy", split_pre_first), ("split, PRE on a side stream enqueued AFTER the policy", split_policy_first)):
    print("whole step  %-62s %.1f" % (name, loop(fn)))
