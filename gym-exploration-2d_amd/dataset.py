"""Trajectory-dataset export (second half).

The reference's dataset creator (experiments/src/run_trajectory_dataset_creator.py:53-109) walks each agent's
`global_state_history` (agent.py:217-232: rows [t, px, py, gx, gy, radius, pref_speed, vx, vy, speed, heading,
a0, a1]) and writes, per agent, a list of records.  Row k is written by Agent.take_action during the agent's
(k+1)-th step AFTER the move and BEFORE `t += dt` (agent.py:176-186, 198-201): position / velocity / heading after
k+1 moves next to the time stamp k * dt; the initial state is never logged, a done agent logs nothing more, and
main() keeps rows [:step_num] (:164-167).  Pinned by tests/golden/adapters.npz (add_traj run on a reference episode).

    {'time', 'pedestrian_goal_position', 'coop_coef', 'other_agents_pos', 'other_agents_vel',
     'pedestrian_state': {'position', 'velocity'}}

This module records the same history for N worlds at once from the batched env (device tensors, one host copy
at the end) and formats it with the same schema.  Files are written with pickle, like the reference's `.pkl`.
"""
import pickle

import numpy as np
import torch

HISTORY_COLUMNS = ("t", "pos_x", "pos_y", "goal_x", "goal_y", "radius", "pref_speed", "vel_x", "vel_y", "speed", "heading")


def record_episode(env, max_steps=1000, actions=None):
    """Step every world of `env` (freshly reset) until every agent is done (no auto-reset) or max_steps.
    Returns dict of numpy arrays: history [T, N, M, 13] in the reference's row convention (row k = state after step k+1,
    time stamp of before it), step_num [N, M] (the rows of agent i are history[:step_num, w, i]), n_agents [N],
    coop [N, M]."""
    rows = []

    def snap():
        st = env.state()
        cols = [st[k] for k in HISTORY_COLUMNS] + [st["action"][..., 0].double(), st["action"][..., 1].double()]
        return torch.stack([c.double() for c in cols], dim=-1)

    prev = snap()
    for _ in range(max_steps):
        env.step(actions)
        cur = snap()
        row = cur.clone()
        row[..., 0] = prev[..., 0]  # Agent._update_state_history runs before `self.t += dt`
        rows.append(row)
        prev = cur
        st = env.state()
        active = torch.arange(env.M, device=env.device)[None, :] < st["n_agents"][:, None]
        if bool((((st["status"] & 8) != 0) | ~active).all()):  # CAGYM_FLAG_DONE for every agent of every world
            break
    torch.cuda.synchronize(env.device)
    st = env.state()
    sc = env.scenarios()
    S = env.S
    sidx = ((torch.arange(env.N, device=env.device) + st["episode"].long() * env.N) % S).long()
    return {"history": torch.stack(rows).cpu().numpy(), "step_num": st["step_num"].cpu().numpy(),
            "n_agents": st["n_agents"].cpu().numpy(), "coop": sc["coop"][sidx].cpu().numpy()}


def to_reference_records(rec, world, dt=0.1, last_time=0.0):
    """add_traj (run_trajectory_dataset_creator.py:53-109) for one world: list (one per agent) of lists of records.
    Returns (trajectories, last_time for the next episode)."""
    H, steps = rec["history"][:, world], rec["step_num"][world]
    n = int(rec["n_agents"][world])
    trajs = []
    d = None
    for i in range(n):
        traj = []
        max_ts = int(steps[i])  # global_state_history[:step_num]: one row per step the agent took
        for t in range(max_ts):
            opos, ovel = [], []
            for j in range(n):
                if j == i:
                    continue
                lj = int(steps[j])
                if t >= lj:  # the other agent finished earlier: last position, zero velocity (:73-75)
                    opos.append((H[lj - 1, j, 1], H[lj - 1, j, 2]))
                    ovel.append((0, 0))
                else:
                    opos.append((H[t, j, 1], H[t, j, 2]))
                    ovel.append((H[t, j, 7], H[t, j, 8]))
            d = {"time": np.round(last_time + t * 0.1, decimals=1),
                 "pedestrian_goal_position": (H[t, i, 3], H[t, i, 4]),
                 "coop_coef": float(rec["coop"][world, i]),
                 "other_agents_pos": opos, "other_agents_vel": ovel,
                 "pedestrian_state": {"position": (H[t, i, 1], H[t, i, 2]), "velocity": (H[t, i, 7], H[t, i, 8])}}
            traj.append(d)
        trajs.append(traj)
    return trajs, (d["time"] + 1.0 if d is not None else last_time)


def export(env, path, max_steps=1000):
    """Record one episode of every world and write the reference-schema trajectories of all of them to `path`."""
    rec = record_episode(env, max_steps)
    out, last = [], 0.0
    for w in range(env.N):
        trajs, last = to_reference_records(rec, w, last_time=last)
        out.extend(trajs)
    with open(path, "wb") as f:
        pickle.dump(out, f)
    return len(out)
