"""ORCA / RVO2 restatement (oracle) -- property tests.  The real rvo2 library is absent ("parity
unpinned"); these are the properties SURVEY lists: no-neighbour -> preferred velocity,
head-on symmetry, speed bound, and goal-reaching without collisions on random worlds."""
import importlib

import numpy as np

from oracle import oracle as orc

scen = importlib.import_module("gym-exploration-2d_amd.scenarios")


def _vel(action, heading):
    sp, dh = action
    h = heading + dh
    return np.array([sp * np.cos(h), sp * np.sin(h)])


def test_no_neighbour_returns_pref_velocity():
    pos = np.array([[0.0, 0.0], [500.0, 500.0]])
    vel = np.zeros((2, 2))
    goal = np.array([[5.0, 0.0], [505.0, 500.0]])
    a = orc.orca_action(pos, vel, goal, [1.0, 1.0], [0.5, 0.5], ego=0, heading=0.0, collab=0.5)
    assert abs(a[0] - 1.0) < 1e-5 and abs(a[1]) < 1e-6


def test_head_on_pair_is_mirror_symmetric():
    pos = np.array([[-3.0, 0.0], [3.0, 0.0]])
    vel = np.array([[1.0, 0.0], [-1.0, 0.0]])
    goal = np.array([[3.0, 0.0], [-3.0, 0.0]])
    a0 = orc.orca_action(pos, vel, goal, [1.0, 1.0], [0.5, 0.5], 0, 0.0, 0.5)
    a1 = orc.orca_action(pos, vel, goal, [1.0, 1.0], [0.5, 0.5], 1, np.pi, 0.5)
    v0, v1 = _vel(a0, 0.0), _vel(a1, np.pi)
    assert np.allclose(v0, -v1, atol=2e-5)
    assert abs(v0[1]) > 1e-3  # they do dodge sideways
    assert np.hypot(*v0) <= 1.0 + 1e-5


def test_speed_bound_and_turn_clamp():
    rng = np.random.default_rng(0)
    for _ in range(200):
        M = rng.integers(2, 11)
        w = scen.random_world(rng, M)
        pos, goal = w[:, 0:2], w[:, 2:4]
        vel = rng.uniform(-1, 1, (M, 2))
        h = rng.uniform(-np.pi, np.pi)
        a = orc.orca_action(pos, vel, goal, w[:, 4], w[:, 5], 0, h, 0.5)
        assert 0.0 <= a[0] <= 1.0 + 1e-4
        assert abs(a[1]) <= np.pi / 6 + 1e-12
        if abs(abs(a[1]) - np.pi / 6) < 1e-12:
            assert a[0] == 0.0  # stop-and-turn clamp (RVOPolicy.py:97-106)


def test_random_worlds_reach_goals_without_collisions():
    N, M = 24, 10
    a6 = scen.random_worlds(N, M, seed=2024)
    env = orc.OracleEnv(N=N, M=M, game_over_mode=orc.GO_ALL)
    env.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5))
    env.reset()
    for _ in range(400):
        env.step()
        if env.u("game_over").all():
            break
    at_goal = env.u("is_at_goal").sum()
    coll = env.u("in_collision").sum()
    assert coll == 0
    assert at_goal >= 0.95 * N * M
