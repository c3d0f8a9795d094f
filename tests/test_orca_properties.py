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


# ---- static obstacles (RVOPolicy.py:56-57; obstacle half of Agent::computeNewVelocity) -- PARITY UNPINNED (rvo2 absent) ----
def _obstacle_world(rng, n_rect, M):
    """2..10 axis-aligned rectangles in the style of test_cases.py:2480-2516 (squares 1-2 m, walls 1-4 x 1-4 m) and agents
    whose starts and goals keep 1 m from every rectangle."""
    rects = []
    for _ in range(n_rect):
        w, h = (rng.uniform(1, 2),) * 2 if rng.uniform() < 0.5 else (rng.uniform(1, 4), rng.uniform(1, 4))
        xu, yu = rng.uniform(-6, 8), rng.uniform(-6, 8)
        rects.append([xu - w, yu - h, xu, yu])
    rects = np.array(rects)

    def free(p, margin=1.0):
        return not ((p[0] > rects[:, 0] - margin) & (p[0] < rects[:, 2] + margin) &
                    (p[1] > rects[:, 1] - margin) & (p[1] < rects[:, 3] + margin)).any()

    a6 = np.zeros((M, 6))
    a6[:, 4], a6[:, 5] = 1.0, 0.5
    for i in range(M):
        while True:
            ang, d = rng.uniform(-np.pi, np.pi), rng.uniform(8, 10)
            s = np.array([d * np.cos(ang), d * np.sin(ang)])
            if free(s) and free(-s) and all(np.hypot(*(s - a6[j, 0:2])) > 1.5 and np.hypot(*(-s - a6[j, 2:4])) > 1.5 for j in range(i)):
                a6[i, 0:2], a6[i, 2:4] = s, -s
                break
    return a6, rects


def test_obstacle_line_known_answer_wall_ahead():
    """An agent 2 m in front of a long wall, moving at it: one obstacle half-plane whose boundary is the wall's cut-off
    line, i.e. the approach speed is capped at (gap - radius) / timeHorizonObst = (2 - 0.575) / 5."""
    pos, vel, goal = np.array([[0., 0.]]), np.array([[1., 0.]]), np.array([[6., 0.]])
    r = orc.orca_action_ex(pos, vel, goal, [1.0], [0.5], 0, 0.0, rects=[[2., -5., 3., 5.]])
    assert r["n_obst_lines"] == 1
    assert np.allclose(r["lines"][0], [0.285, 1.0, 0.0, 1.0], atol=1e-6)  # point (0.285, *), direction +y: feasible side v_x <= 0.285
    assert np.allclose(r["new_vel"], [0.285, 0.0], atol=1e-6)
    # a wall behind the agent (it is on the wall's right side, but moving away): the line exists and does not bind
    r = orc.orca_action_ex(pos, vel, goal, [1.0], [0.5], 0, 0.0, rects=[[-3., -5., -2., 5.]])
    assert r["n_obst_lines"] == 1 and np.allclose(r["new_vel"], [1.0, 0.0], atol=1e-6)
    # out of range (timeHorizonObst * maxSpeed + radius = 5.575 m): no line
    r = orc.orca_action_ex(pos, vel, goal, [1.0], [0.5], 0, 0.0, rects=[[6., -5., 7., 5.]])
    assert r["n_obst_lines"] == 0


def test_free_space_unchanged_by_the_obstacle_code():
    """Rectangles out of every agent's range leave the solve bit for bit as it was (and so does an empty list)."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        M = int(rng.integers(2, 11))
        w = scen.random_world(rng, M)
        vel = rng.uniform(-1, 1, (M, 2))
        h = rng.uniform(-np.pi, np.pi)
        base = orc.orca_action(w[:, 0:2], vel, w[:, 2:4], w[:, 4], w[:, 5], 0, h, 0.5)
        far = orc.orca_action_ex(w[:, 0:2], vel, w[:, 2:4], w[:, 4], w[:, 5], 0, h, 0.5, max_neighbors=10,
                                 rects=[[40., 40., 42., 43.], [-50., 10., -48., 11.]])
        none = orc.orca_action_ex(w[:, 0:2], vel, w[:, 2:4], w[:, 4], w[:, 5], 0, h, 0.5, max_neighbors=10, rects=None)
        assert np.array_equal(base, far["action"]) and np.array_equal(base, none["action"]) and far["n_obst_lines"] == 0


def test_obstacle_lines_are_satisfied_and_mirror_symmetric():
    """Whenever linearProgram2 is feasible the chosen velocity satisfies every half-plane (obstacle lines first); and the
    mirror image of a scene (y -> -y) yields the mirror-image velocity."""
    rng = np.random.default_rng(11)
    checked = 0
    for _ in range(300):
        M = int(rng.integers(1, 6))
        a6, rects = _obstacle_world(rng, int(rng.integers(2, 8)), M)
        pos = a6[:, 0:2] * rng.uniform(0.2, 1.0)  # somewhere along the way, possibly close to rectangles
        inside = ((pos[0, 0] > rects[:, 0] - 0.6) & (pos[0, 0] < rects[:, 2] + 0.6) & (pos[0, 1] > rects[:, 1] - 0.6) & (pos[0, 1] < rects[:, 3] + 0.6)).any()
        if inside:
            continue
        vel = rng.uniform(-1, 1, (M, 2))
        r = orc.orca_action_ex(pos, vel, a6[:, 2:4], a6[:, 4], a6[:, 5], 0, 0.3, 0.5, rects=rects)
        v, L = r["new_vel"].astype(np.float64), r["lines"].astype(np.float64)
        assert np.hypot(*v) <= 1.0 + 5e-4  # fp32: a point on the disc boundary computed along a far-away line
        viol = L[:, 2] * (L[:, 1] - v[1]) - L[:, 3] * (L[:, 0] - v[0]) if len(L) else np.zeros(0)  # det(dir, point - v) <= 0
        if len(L) and viol.max() <= 1e-4:
            checked += 1
        # obstacle lines are hard constraints even when the agent lines are infeasible (linearProgram3 keeps them)
        no = r["n_obst_lines"]
        if no and not inside:
            assert viol[:no].max() <= 2e-4 or viol.max() > 1e-4
        fl = np.array([1.0, -1.0])
        rm = rects[:, [0, 3, 2, 1]] * np.array([1, -1, 1, -1])
        m = orc.orca_action_ex(pos * fl, vel * fl, a6[:, 2:4] * fl, a6[:, 4], a6[:, 5], 0, -0.3, 0.5, rects=rm)
        assert m["n_obst_lines"] == no
        assert np.allclose(m["new_vel"] * fl, r["new_vel"], atol=5e-5)
    assert checked > 100


def test_rvo_agents_among_obstacles_do_not_hit_walls():
    """Lone RVO agents crossing fields of rectangles (no other agents in the way): never a wall collision, most arrive."""
    rng = np.random.default_rng(2)
    N, M, K = 32, 10, 10
    a6 = np.zeros((N, M, 6))
    obst = np.zeros((N, K, 4))
    n_obst = np.zeros(N, dtype=np.int32)
    for w in range(N):
        n_obst[w] = rng.integers(2, K + 1)
        a, r = _obstacle_world(rng, n_obst[w], 1)
        a6[w, 0], obst[w, :n_obst[w]] = a[0], r
        a6[w, 1:, 4], a6[w, 1:, 5] = 1.0, 0.5
    env = orc.OracleEnv(N=N, M=M, max_obstacles=K, game_over_mode=orc.GO_ALL)
    env.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, n_agents=np.ones(N, dtype=np.int32), coop=np.full((N, M), 0.5),
                     obstacles=obst, n_obst=n_obst)
    env.reset()
    for _ in range(400):
        env.step()
        if env.u("game_over").all():
            break
    assert env.u("in_collision")[:, 0].sum() == 0
    assert env.u("is_at_goal")[:, 0].mean() >= 0.6  # ORCA is not a planner: some agents stay stuck behind a wall and time out


def test_rvo_crowds_among_obstacles_mostly_arrive():
    rng = np.random.default_rng(8)
    N, M, K = 16, 6, 8
    a6 = np.zeros((N, M, 6))
    obst = np.zeros((N, K, 4))
    n_obst = np.zeros(N, dtype=np.int32)
    for w in range(N):
        n_obst[w] = rng.integers(2, K + 1)
        a6[w], r = _obstacle_world(rng, n_obst[w], M)
        obst[w, :n_obst[w]] = r
    env = orc.OracleEnv(N=N, M=M, max_obstacles=K, game_over_mode=orc.GO_ALL)
    env.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5), obstacles=obst, n_obst=n_obst)
    env.reset()
    for _ in range(500):
        env.step()
        if env.u("game_over").all():
            break
    n = N * M
    assert env.u("in_collision").sum() <= 0.05 * n
    assert env.u("is_at_goal").sum() >= 0.5 * n
