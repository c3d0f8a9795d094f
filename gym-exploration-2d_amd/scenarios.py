"""Host-side scenario arrays for the batched env.

The reference builds scenarios as lists of Agent objects with np.random rejection
sampling (test_cases.py:1192-1463) or from "legacy cadrl" arrays
[start_x, start_y, goal_x, goal_y, pref_speed, radius] (test_cases.py:1970-2014).
The batched env consumes the array form directly: agents6[N, M, 6] float64.

`random_worlds` implements the synthetic rule of SURVEY.md section 8(d) (itself the
rule of test_cases.py:1392-1424): start, goal ~ U([-7.5, 7.5]^2), |goal-start| >= 4,
pairwise start/start and goal/goal separation >= 1.5 m, radius 0.5, pref_speed 1.0.
"""
import numpy as np

# policy ids (policies/*.py of the reference; see include/cagym.h)
POLICY_STATIC = 0        # StaticPolicy.py:9-12
POLICY_NONCOOP = 1       # NonCooperativePolicy.py:10-13
POLICY_EXTERNAL = 2      # raw (speed, delta_heading) pair
POLICY_LEARNING = 3      # LearningPolicy.py:11-16
POLICY_CARRL = 4         # CARRLPolicy.py:5-15 (11-row discrete table)
POLICY_RVO = 5           # RVOPolicy.py:53-117 (ORCA)
POLICY_GA3C = 6          # GA3CCADRLPolicy.py:34-43
POLICY_IGMCTS = 7        # ig_mcts.py (planner supplies (v, omega) as external action)

# dynamics ids (dynamics/*.py of the reference)
DYN_UNICYCLE = 0         # UnicycleDynamics.py:10-31
DYN_MAXTURNRATE = 1      # UnicycleDynamicsMaxTurnRate.py:11-25
DYN_MAXACC = 2           # UnicycleDynamicsMaxAcc.py:17-39
DYN_SECONDORDER = 3      # UnicycleSecondOrderEulerDynamics.py:12-29
DYN_FIRSTORDER = 4       # FirstOrderDynamics.py:10-23


def random_world(rng, M, side=7.5, min_travel=4.0, min_sep=1.5, radius=0.5, pref_speed=1.0):
    """One world: float64 [M, 6] rows [sx, sy, gx, gy, pref_speed, radius]."""
    out = np.zeros((M, 6), dtype=np.float64)
    starts, goals = [], []
    for i in range(M):
        while True:
            s = rng.uniform(-side, side, 2)
            g = rng.uniform(-side, side, 2)
            if np.hypot(*(g - s)) < min_travel:
                continue
            if any(np.hypot(*(s - q)) < min_sep for q in starts):
                continue
            if any(np.hypot(*(g - q)) < min_sep for q in goals):
                continue
            break
        starts.append(s)
        goals.append(g)
        out[i] = [s[0], s[1], g[0], g[1], pref_speed, radius]
    return out


def random_worlds(N, M, seed=1234, **kw):
    """agents6[N, M, 6]; world w uses default_rng(seed + w)."""
    return np.stack([random_world(np.random.default_rng(seed + w), M, **kw) for w in range(N)])


def random_worlds_fast(N, M, seed=1234, side=7.5, min_travel=4.0, min_sep=1.5,
                       radius=0.5, pref_speed=1.0):
    """Vectorised rejection sampler for large pools (bench): same constraints as
    `random_world`, different (batched) RNG stream."""
    rng = np.random.default_rng(seed)
    out = np.zeros((N, M, 6), dtype=np.float64)
    out[:, :, 4] = pref_speed
    out[:, :, 5] = radius
    for i in range(M):
        todo = np.arange(N)
        while todo.size:
            s = rng.uniform(-side, side, (todo.size, 2))
            g = rng.uniform(-side, side, (todo.size, 2))
            ok = np.hypot(g[:, 0] - s[:, 0], g[:, 1] - s[:, 1]) >= min_travel
            for j in range(i):
                ps = out[todo, j, 0:2]
                pg = out[todo, j, 2:4]
                ok &= np.hypot(s[:, 0] - ps[:, 0], s[:, 1] - ps[:, 1]) >= min_sep
                ok &= np.hypot(g[:, 0] - pg[:, 0], g[:, 1] - pg[:, 1]) >= min_sep
            w = todo[ok]
            out[w, i, 0:2] = s[ok]
            out[w, i, 2:4] = g[ok]
            todo = todo[~ok]
    return out


def circle_world(M, radius):
    """gen_circle_test_case (test_cases.py:2207-2218): antipodal swap on a circle."""
    tc = np.zeros((M, 6))
    for i in range(M):
        th0 = (2 * np.pi / M) * i
        th1 = th0 + np.pi
        tc[i] = [radius * np.cos(th0), radius * np.sin(th0),
                 radius * np.cos(th1), radius * np.sin(th1), 1.0, 0.5]
    return tc


def heading_toward_goal(agents6):
    """Initial heading when none is given (agent.py:29-31)."""
    a = np.asarray(agents6, dtype=np.float64)
    return np.arctan2(a[..., 3] - a[..., 1], a[..., 2] - a[..., 0])


def obstacle_worlds(N, M, K, seed=1234, n_agents_min=None):
    """Worlds in the style of train_stage_2 (test_cases.py:2464-2572), vectorised: per world 2..K non-overlapping
    axis-aligned rectangles (square of side U(1,2), or wall U(1,4) x (U(1,2) if wide else U(3,4)); upper corner U(-8,10)^2;
    is_shape_valid :150-170), every agent starts at distance U(8,10) from the origin at a random angle with its goal at
    the antipode, start and goal at least 1 m from every rectangle (is_pose_valid_with_obstacles :135-148) and 1.5 m
    from every earlier start / goal (is_pose_valid :129-133); radius 0.5, pref_speed 1.  A different (batched) RNG stream
    than the reference's.  Returns agents6 [N, M, 6], obstacles [N, K, 4] (xl, yl, xu, yu), n_obst [N], n_agents [N]
    (M everywhere unless n_agents_min is given: U{n_agents_min..M} as :2534)."""
    rng = np.random.default_rng(seed)
    n_obst = rng.integers(2, K + 1, N).astype(np.int32)
    obst = np.zeros((N, K, 4))
    for k in range(K):
        todo = np.nonzero(n_obst > k)[0]
        tries = 0
        while todo.size:
            tries += 1
            if tries > 200:  # no room left for another rectangle in these worlds: they keep the k they have
                n_obst[todo] = k
                break
            n = todo.size
            square = rng.uniform(size=n) < 0.5
            sq = rng.uniform(1, 2, n)
            wx = rng.uniform(1, 4, n)
            wy = np.where(wx > 2, rng.uniform(1, 2, n), rng.uniform(3, 4, n))
            sx, sy = np.where(square, sq, wx), np.where(square, sq, wy)
            xu, yu = rng.uniform(-8, 10, n), rng.uniform(-8, 10, n)
            xl, yl = xu - sx, yu - sy
            ok = np.ones(n, dtype=bool)
            for j in range(k):
                o = obst[todo, j]
                ok &= (o[:, 0] >= xu) | (xl >= o[:, 2]) | (o[:, 3] <= yl) | (yu <= o[:, 1])
            w = todo[ok]
            obst[w, k] = np.stack([xl[ok], yl[ok], xu[ok], yu[ok]], 1)
            todo = todo[~ok]
    a6 = np.zeros((N, M, 6))
    a6[:, :, 4], a6[:, :, 5] = 1.0, 0.5

    def clear(p, idx):  # 1 m from every rectangle of the world
        o = obst[idx]
        live = np.arange(K)[None, :] < n_obst[idx, None]
        inside = (p[:, None, 0] < o[:, :, 2] + 1) & (p[:, None, 1] < o[:, :, 3] + 1) & (p[:, None, 0] > o[:, :, 0] - 1) & (p[:, None, 1] > o[:, :, 1] - 1)
        return ~(inside & live).any(1)

    for i in range(M):
        todo = np.arange(N)
        tries = 0
        while todo.size:
            tries += 1
            if tries % 200 == 0:  # rectangles leave no admissible start/goal pair: such a world gives up its last rectangle
                n_obst[todo] = np.maximum(n_obst[todo] - 1, 0)
            if tries > 200 * (K + 2):
                raise RuntimeError("obstacle_worlds: could not place agent %d in %d worlds" % (i, todo.size))
            n = todo.size
            d, ang = rng.uniform(8, 10, n), rng.uniform(-np.pi, np.pi, n)
            s = np.stack([d * np.cos(ang), d * np.sin(ang)], 1)
            ok = clear(s, todo) & clear(-s, todo)
            for j in range(i):
                ok &= np.hypot(*(s - a6[todo, j, 0:2]).T) >= 1.5
                ok &= np.hypot(*(s - a6[todo, j, 2:4]).T) >= 1.5
                ok &= np.hypot(*(-s - a6[todo, j, 0:2]).T) >= 1.5
                ok &= np.hypot(*(-s - a6[todo, j, 2:4]).T) >= 1.5
            w = todo[ok]
            a6[w, i, 0:2], a6[w, i, 2:4] = s[ok], -s[ok]
            todo = todo[~ok]
    obst[np.arange(K)[None, :] >= n_obst[:, None]] = 0.0
    n_agents = np.full(N, M, dtype=np.int32) if n_agents_min is None else rng.integers(n_agents_min, M + 1, N).astype(np.int32)
    return a6, obst, n_obst, n_agents
