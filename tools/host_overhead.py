"""Host-side cost of one bracketed launch (what bench.py's wall clock adds to the kernel time of a 20-step block): micro-timings
of the pieces of BatchedCollisionAvoidanceEnv.rollout and of the bracket itself, on a tiny handle."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv


def per_call(fn, n=20000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


N, M = 64, 10
env = B(N, M, game_over_mode="all")
env.set_scenarios(scen.random_worlds_fast(N, M, seed=1), scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5))
env.reset()
traj = env.alloc_rollout(20)
dev = env.device


def ctx():
    with torch.cuda.device(dev):
        pass


print("with torch.cuda.device(dev): pass      %6.2f us" % per_call(ctx))
print("env._stream()                          %6.2f us" % per_call(env._stream))
print("env._outputs(6 tensors)                %6.2f us" % per_call(lambda: env._outputs(traj.get("other_agents_states"), traj.get("ego"), traj.get("laserscan"), traj.get("reward"), traj.get("flags"), traj.get("game_over"))))
print("ctypes call cagym_version()            %6.2f us" % per_call(env.L.cagym_version))
ev = torch.cuda.Event(enable_timing=True)
print("event.record()                         %6.2f us" % per_call(ev.record, 2000))
torch.cuda.synchronize()
print("torch.cuda.synchronize() (idle)        %6.2f us" % per_call(lambda: torch.cuda.synchronize(dev), 2000))


def bracket(with_events):
    torch.cuda.synchronize(dev)
    if with_events:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if with_events:
        e0.record()
    env.rollout(20, out=traj)
    if with_events:
        e1.record()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    return el * 1e6, (e0.elapsed_time(e1) * 1e3 if with_events else 0.0)


for we in (False, True):
    r = sorted(bracket(we) for _ in range(500))
    print("bracketed 20-step rollout of 64 worlds, events %-5s: wall median %7.2f us, HIP events %7.2f us" % (we, r[250][0], sorted(x[1] for x in r)[250]))
env.close()
