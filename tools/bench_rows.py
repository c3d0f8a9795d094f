#!/usr/bin/env python3
"""Secondary measurements for the non-headline rows of SURVEY.md 8(a) (MI355X, one GPU):
  cfg4-style: 8192 worlds x 10 agents with obstacles, agent 0 GA3C-CADRL (+LaserScan on all agents), 9 RVO
  cfg5-style: information-gain primitives -- visibility queries/s, belief updates/s, roll-outs/s
Prints one JSON object; numbers are recorded in DESIGN.md / profiles/."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
IG = importlib.import_module("gym-exploration-2d_amd.ig").InfoGain


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def rects(rng, S, K):
    ob = np.zeros((S, K, 4))
    n = rng.integers(2, K + 1, S).astype(np.int32)
    c = rng.uniform(-12, 12, (S, K, 2))
    h = rng.uniform(0.3, 1.5, (S, K, 2))
    ob[..., 0], ob[..., 1], ob[..., 2], ob[..., 3] = c[..., 0] - h[..., 0], c[..., 1] - h[..., 1], c[..., 0] + h[..., 0], c[..., 1] + h[..., 1]
    return ob, n


out = {}
rng = np.random.default_rng(0)
# ---- cfg4-style -------------------------------------------------------------------------------------------
N, M, K = 8192, 10, 10
a6 = scen.random_worlds_fast(N, M, seed=3)
pol = np.full((N, M), scen.POLICY_RVO, dtype=np.int32)
pol[:, 0] = scen.POLICY_GA3C
ob, nob = rects(rng, N, K)
env = B(N, M, max_obstacles=K, laserscan=True, game_over_mode="agent0")
env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5), obstacles=ob, n_obst=nob)
env.reset()
policy = GA3C(env)
ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)


def step4():
    policy.act(ext)
    _, _, go, _ = env.step(ext)
    env.reset(world_mask=go)


dt = timed(step4, 50)
out["cfg4_8192x10_ga3c_agent0_rvo9_laserscan_obstacles"] = {"ms_per_step": dt * 1e3, "env_steps_per_s": N / dt,
                                                           "note": "GA3C state kernel + fused forward kernel (8192 agents) + cagym_step + laserscan + masked reset, per-step launches"}
dt_nn = timed(lambda: policy.act(ext), 50)
out["ga3c_state_plus_forward_8192_agents_ms"] = dt_nn * 1e3
dt_ls = timed(lambda: env.sense_laserscan(), 50)
out["laserscan_81920_agents_ms"] = dt_ls * 1e3
env.close()
# second cfg4 run of SURVEY 8(d): GA3C + LaserScan on ALL agents
pol_all = np.full((N, M), scen.POLICY_GA3C, dtype=np.int32)
env = B(N, M, max_obstacles=K, laserscan=True, game_over_mode="agent0")
env.set_scenarios(a6, pol_all, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5), obstacles=ob, n_obst=nob)
env.reset()
policy = GA3C(env)
dt = timed(step4, 30)
out["cfg4b_8192x10_ga3c_all_agents_laserscan_obstacles"] = {"ms_per_step": dt * 1e3, "env_steps_per_s": N / dt,
                                                           "agent_nn_evals_per_s": N * M / dt}
env.close()
# ---- cfg5-style -------------------------------------------------------------------------------------------
N, M = 2048, 20
a6 = scen.random_worlds_fast(N, M, seed=4)
ob, nob = rects(rng, N, 8)
env = B(N, M, max_obstacles=8, game_over_mode="all")
env.set_scenarios(a6, scen.POLICY_NONCOOP, scen.DYN_FIRSTORDER, obstacles=ob, n_obst=nob)
env.reset()
t0 = time.perf_counter()
ig = IG(env)
torch.cuda.synchronize()
out["edt_2048_scenarios_ms"] = (time.perf_counter() - t0) * 1e3
Q = 2048 * 32
poses = torch.from_numpy(np.concatenate([rng.uniform(-12, 12, (Q, 2)), rng.uniform(-np.pi, np.pi, (Q, 1))], 1)).to(env.device)
world = torch.arange(Q, device=env.device, dtype=torch.int32) % N
dt = timed(lambda: ig.visible_cells(poses, world), 10)
out["visibility_queries_per_s"] = Q / dt
masks = ig.visible_cells(poses, world)
dt = timed(lambda: ig.mi_reward(masks, world), 10)
out["mi_rewards_per_s"] = Q / dt
P3 = poses[:N * 3].reshape(N, 3, 3)
det = torch.zeros((N, 3, 2, 2), dtype=torch.float64, device=env.device)
nd = torch.zeros((N, 3), dtype=torch.int32, device=env.device)
dt = timed(lambda: ig.update_belief(P3, det, nd), 10)
out["belief_updates_per_s_3_poses_per_world"] = N / dt
Qr, nsims, H = 2048 * 3, 10, 4  # exp/dmcts.py budget: Nsims 10, horizon 4, xdt 5
zeros = torch.zeros((Qr, 60), dtype=torch.int64, device=env.device)
dt = timed(lambda: ig.rollouts(poses[:Qr], zeros, zeros, world[:Qr], torch.full((Qr,), H), torch.full((Qr,), 0.5), nsims, 7), 10)
out["rollouts_per_s_horizon4"] = Qr * nsims / dt
out["rollout_visibility_queries_per_s"] = Qr * nsims * H / dt
env.close()
# cfg5 env part: 2048 worlds x 20 agents = 3 IG agents (external (v, omega), FirstOrderDynamics) + 2 static
# targets + 15 NonCooperative, obstacles, OAS [19, 10] observations; per-step launches with external actions
pol5 = np.full((N, M), scen.POLICY_NONCOOP, dtype=np.int32)
pol5[:, :3] = scen.POLICY_IGMCTS
pol5[:, 3:5] = scen.POLICY_STATIC
dyn5 = np.full((N, M), scen.DYN_UNICYCLE, dtype=np.int32)
dyn5[:, :3] = scen.DYN_FIRSTORDER
env = B(N, M, max_obstacles=8, game_over_mode="all")
env.set_scenarios(a6, pol5, dyn5, obstacles=ob, n_obst=nob)
env.reset()
ext5 = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
ext5[:, :3, 0] = 2.0
dt = timed(lambda: env.step(ext5, auto_reset=True), 100)
out["cfg5_env_part_2048x20"] = {"ms_per_step": dt * 1e3, "env_steps_per_s": N / dt, "agent_steps_per_s": N * M / dt,
                                "hbm_frac_917B_per_agent_step": N * M / dt * 917.0 / 8e12}
# Dec-MCTS planning step for 3 robots per world on 64 worlds, exp/dmcts.py budget (Ntree 30, Nsims 10, horizon 4,
# Ncycles 5): host tree + device primitives
dm = importlib.import_module("gym-exploration-2d_amd.dmcts")
igm = importlib.import_module("gym-exploration-2d_amd.ig")
ig = IG(env)
NW = 64
planner = dm.DecMCTSPlanner(igm.InfoGainBackend(ig), NW, 3, radius=0.5, Ntree=30, Nsims=10, horizon=4, Ncycles=5, seed=1)
pp = np.concatenate([a6[:NW, :3, 0:2], np.zeros((NW, 3, 1))], axis=2)
t0 = time.perf_counter()
planner.plan(pp)
dtp = time.perf_counter() - t0
out["dmcts_plan_64_worlds_3_robots_s"] = dtp
out["dmcts_rollouts_per_s_incl_host_tree"] = NW * 3 * 5 * 30 * 10 / dtp
# the same planning step with the trees on the device, for ALL 2048 worlds
dplanner = dm.DeviceDecMCTSPlanner(ig, 3, radius=0.5, Ntree=30, Nsims=10, horizon=4, Ncycles=5, seed=1)
pp_all = torch.from_numpy(np.concatenate([a6[:, :3, 0:2], np.zeros((N, 3, 1))], axis=2)).to(env.device)
dplanner.plan(pp_all)
torch.cuda.synchronize()
t0 = time.perf_counter()
dplanner.plan(pp_all)
torch.cuda.synchronize()
dtd = time.perf_counter() - t0
out["dmcts_device_plan_2048_worlds_3_robots_s"] = dtd
out["dmcts_device_rollouts_per_s"] = N * 3 * 5 * 30 * 10 / dtd
out["dmcts_device_workspace_GB"] = dplanner.workspace.numel() / 1e9
env.close()
print(json.dumps(out, indent=1))
