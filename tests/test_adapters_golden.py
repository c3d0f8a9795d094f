"""The two adapters either side of the hot path against fixtures produced by the reference's own code
(tests/golden/adapters.npz, made by tests/golden/make_golden.py --adapters-only):
  * vecenv.FlatObservation  vs  MultiagentFlattenDictWrapper.observation (envs/wrappers.py:38-46) on every step of a
    reference episode (4 agents in a 10-slot env: the empty slots are part of the 1010-float vector);
  * dataset.to_reference_records  vs  add_traj (experiments/src/run_trajectory_dataset_creator.py:53-109) on the same episode."""
import importlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "adapters.npz")
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")


def _env(z, M=10):
    B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    n = z["agents6"].shape[0]
    a6 = np.zeros((1, M, 6))
    a6[0, :, 4], a6[0, :, 5] = 1.0, 0.1
    a6[0, n:, 0] = 1e3 + np.arange(M - n)
    a6[0, :n] = z["agents6"]
    pol = np.zeros((1, M), dtype=np.int32)
    pol[0, :n] = z["policy_id"]
    h0 = np.zeros((1, M))
    h0[0, :n] = z["heading0"]
    coop = np.ones((1, M))
    coop[0, :n] = z["coop"]
    env = B(1, M, game_over_mode="all")
    env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, heading0=h0, n_agents=[n], coop=coop)
    env.reset()
    return env, n


def test_fixture_layout_matches_the_wrapper_bookkeeping():
    """observation_indices of the adapter == MultiagentFlattenDictWrapper.observation_indices (wrappers.py:17-31)."""
    vec = importlib.import_module("gym-exploration-2d_amd.vecenv")
    z = np.load(GOLD)
    keys = [str(k) for k in z["keys"]]
    idx, size = vec.observation_indices(keys, 10)
    assert size == z["flat"].shape[1] == 1010
    for k in keys:
        assert np.array_equal(np.array([idx[a][k] for a in range(10)]), z["idx__" + k]), k


@pytest.mark.gpu
def test_flat_observation_matches_reference_wrapper():
    vec = importlib.import_module("gym-exploration-2d_amd.vecenv")
    z = np.load(GOLD)
    keys = [str(k) for k in z["keys"]]
    env, n = _env(z)
    flat = vec.FlatObservation(env, keys)
    T = z["flat"].shape[0]
    worst = 0.0
    for t in range(T):
        if t:
            env.step()
        got = flat().double().cpu().numpy()[0]
        exp = z["flat"][t]
        worst = max(worst, float(np.abs(got - exp).max()))
        assert np.abs(got - exp).max() <= 2e-5, (t, int(np.abs(got - exp).argmax()))  # fp32 observation tensors
    assert np.count_nonzero(z["flat"][0][404:]) == 0 and np.count_nonzero(flat().cpu().numpy()[0][404:]) == 0  # empty slots
    env.close()


@pytest.mark.gpu
def test_dataset_records_match_reference_add_traj():
    ds = importlib.import_module("gym-exploration-2d_amd.dataset")
    z = np.load(GOLD)
    env, n = _env(z)
    rec = ds.record_episode(env, max_steps=1000)
    assert np.array_equal(rec["step_num"][0, :n], z["step_num"])
    trajs, last = ds.to_reference_records(rec, world=0, last_time=5.0)
    assert len(trajs) == int(z["n_traj"]) and abs(last - float(z["last_time"])) < 1e-12
    for i, tr in enumerate(trajs):
        g = lambda k: z["traj%d__%s" % (i, k)]
        assert len(tr) == len(g("time"))
        assert np.array_equal(np.array([d["time"] for d in tr]), g("time"))
        assert np.abs(np.array([d["pedestrian_goal_position"] for d in tr]) - g("goal")).max() == 0.0
        assert np.array_equal(np.array([d["coop_coef"] for d in tr]), g("coop"))
        assert np.abs(np.array([d["pedestrian_state"]["position"] for d in tr]) - g("pos")).max() <= 1e-9
        assert np.abs(np.array([d["pedestrian_state"]["velocity"] for d in tr]) - g("vel")).max() <= 1e-9
        assert np.abs(np.array([d["other_agents_pos"] for d in tr], dtype=np.float64) - g("other_pos")).max() <= 1e-9
        assert np.abs(np.array([d["other_agents_vel"] for d in tr], dtype=np.float64) - g("other_vel")).max() <= 1e-9
    env.close()
