"""Trajectory-dataset export (N4): schema of run_trajectory_dataset_creator.py:53-109 from batched device histories."""
import importlib
import pickle

import numpy as np
import pytest

scen = importlib.import_module("gym-exploration-2d_amd.scenarios")


@pytest.mark.gpu
def test_export_schema_and_consistency(tmp_path):
    B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    ds = importlib.import_module("gym-exploration-2d_amd.dataset")
    N, M = 6, 4
    env = B(N, M, game_over_mode="all")
    na = np.array([4, 3, 4, 2, 4, 4], dtype=np.int32)
    env.set_scenarios(scen.random_worlds_fast(N, M, seed=3), scen.POLICY_RVO, scen.DYN_UNICYCLE, n_agents=na, coop=np.full((N, M), 0.5))
    env.reset()
    rec = ds.record_episode(env, max_steps=400)
    H, steps = rec["history"], rec["step_num"]
    assert H.shape[1:] == (N, M, 13) and (steps.max(axis=1) > 10).all()
    trajs, last = ds.to_reference_records(rec, world=1)
    assert len(trajs) == 3  # one trajectory per agent of the world
    for i, traj in enumerate(trajs):
        assert len(traj) == steps[1, i]  # global_state_history[:step_num] (run_trajectory_dataset_creator.py:164-167)
        r0, r1 = traj[0], traj[-1]
        assert set(r0) == {"time", "pedestrian_goal_position", "coop_coef", "other_agents_pos", "other_agents_vel", "pedestrian_state"}
        assert r0["time"] == 0.0 and abs(r1["time"] - 0.1 * (steps[1, i] - 1)) < 1e-9 and r0["coop_coef"] == 0.5
        assert len(r0["other_agents_pos"]) == 2 and len(r0["other_agents_vel"]) == 2
        # consecutive positions differ by velocity * dt (unicycle: the stored velocity is the one that produced the move)
        for a, b in zip(traj[:-1], traj[1:]):
            p0, p1, v1 = np.array(a["pedestrian_state"]["position"]), np.array(b["pedestrian_state"]["position"]), np.array(b["pedestrian_state"]["velocity"])
            assert np.abs(p1 - p0 - 0.1 * v1).max() < 1e-12
        # reached its goal (ORCA in free space), and the goal is the scenario's
        assert np.hypot(*(np.array(r1["pedestrian_state"]["position"]) - np.array(r1["pedestrian_goal_position"]))) <= 0.75 + 1e-9
    # an agent that finished earlier is reported at its last position with zero velocity
    slow, fast = int(np.argmax(steps[1, :3])), int(np.argmin(steps[1, :3]))
    if steps[1, slow] > steps[1, fast]:
        rec_late = trajs[slow][-1]
        k = [j for j in range(3) if j != slow].index(fast)
        assert rec_late["other_agents_vel"][k] == (0, 0)
        assert rec_late["other_agents_pos"][k] == tuple(trajs[fast][-1]["pedestrian_state"]["position"])
    assert last == trajs[-1][-1]["time"] + 1.0
    path = tmp_path / "trajs.pkl"
    env.reset()
    n = ds.export(env, str(path), max_steps=400)
    assert n == int(na.sum())
    with open(path, "rb") as f:  # a file this test wrote itself
        back = pickle.load(f)
    assert len(back) == n and "pedestrian_state" in back[0][0]
    env.close()
