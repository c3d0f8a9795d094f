"""Replay the reference-generated golden episodes (tests/golden/*.npz) through a backend.

A backend exposes set_scenario / reset / step / f(name) / u(name) / i(name) for N worlds x M
slots: oracle.oracle.OracleEnv (CPU restatement) and the HIP env's state views both do.
"""
import glob
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FLOAT_KEYS = ["pos", "vel", "heading", "speed", "delta_heading", "dist_to_goal", "past_dist_to_goal",
              "heading_ego", "vel_ego", "ref_prll", "rel_goal", "time_remaining", "t", "past_actions", "reward"]
ANGLE_KEYS = ("heading", "heading_ego", "delta_heading")
MASK_KEYS = ["is_at_goal", "was_at_goal_already", "in_collision", "was_in_collision_already",
             "ran_out_of_time", "is_done"]


def load_cases(group):
    z = np.load(os.path.join(GOLD, group + ".npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    return {n: {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(n + "__")} for n in names}


def all_groups():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))
                  if not os.path.basename(p).startswith(("ig_", "ga3c_states", "scenario_", "adapters")))  # episode fixtures only


def game_over_mode(cfg):
    evaluate, homogeneous, single = int(cfg[0]), int(cfg[1]), int(cfg[2])
    if evaluate:
        return 1 if homogeneous else 0
    return 0 if single else 2


def margin_audit(case):
    """min |distance - threshold| over the episode for the three mask-defining tests
    (SURVEY section 7 'Hard parts'): pair collision, goal radius, timeout."""
    pos, a6 = case["pos"], case["agents6"]
    r = a6[:, 5]
    M = pos.shape[1]
    m_coll = np.inf
    for i in range(M):
        for j in range(i + 1, M):
            d = np.hypot(*(pos[:, i] - pos[:, j]).T)
            m_coll = min(m_coll, np.abs(d - (r[i] + r[j])).min())
    dg = np.hypot(*(pos - a6[None, :, 2:4]).transpose(2, 0, 1))
    moving = ~case["is_done"]
    m_goal = np.abs(dg - 0.75)[moving].min() if moving.any() else np.inf
    m_time = np.abs(case["time_remaining"]).min()
    return m_coll, m_goal, m_time


def canon_oas(oas, n_obs, tie=1e-9):
    """Order-canonicalise OAS rows whose sort keys (column 8, edge distance) tie within `tie`.
    The reference's order among such rows depends on last-ulp libm/BLAS rounding of positions
    (symmetric presets: circle, cross), i.e. it is ill-conditioned; rows whose keys are separated
    by more than `tie` keep their order and are therefore compared position by position."""
    out = np.array(oas, dtype=np.float64, copy=True)
    ties = 0
    for i in range(out.shape[0]):
        n = int(n_obs[i])
        k = 0
        while k < n:
            j = k + 1
            while j < n and abs(out[i, j, 8] - out[i, j - 1, 8]) <= tie:
                j += 1
            if j - k > 1:
                blk = out[i, k:j]
                order = np.lexsort((np.round(blk[:, 1], 6), np.round(blk[:, 0], 6)))
                out[i, k:j] = blk[order]
                ties += 1
            k = j
    return out, ties


def oas_mismatch(a, b, n_obs, tol=1e-5, key_tie=1e-5):
    """Largest abs difference between two worlds' OAS tables [M, K, 10] after matching every row of `a` to the row of `b`
    that describes the same other agent (nearest (dx, dy)).  Rows may only sit at different positions where their
    sort keys (column 8) tie within `key_tie`: the order among such rows is decided by last-ulp rounding.  Returns
    (max abs difference, number of displaced rows); raises AssertionError on an unexplained displacement."""
    worst, moved = 0.0, 0
    for i in range(a.shape[0]):
        n = int(n_obs[i])
        if n == 0:
            worst = max(worst, float(np.abs(a[i] - b[i]).max()))
            continue
        A, B = np.asarray(a[i, :n], dtype=np.float64), np.asarray(b[i, :n], dtype=np.float64)
        d = np.abs(A[:, None, 0] - B[None, :, 0]) + np.abs(A[:, None, 1] - B[None, :, 1])
        p = d.argmin(axis=1)
        assert len(set(p.tolist())) == n, "rows do not describe the same set of agents"
        worst = max(worst, float(np.abs(A - B[p]).max()), float(np.abs(a[i, n:] - b[i, n:]).max()) if n < a.shape[1] else 0.0)
        for r in range(n):
            if p[r] != r:
                moved += 1
                assert abs(B[p[r], 8] - B[r, 8]) <= key_tie, "row %d of agent %d displaced without a key tie" % (r, i)
    return worst, moved


def replay(case, make_env, ftol=1e-12, oas_tol=1e-12, laser_tol=1e-12, check=None, float_keys=None,
           tie=1e-9, reward_tol=None):
    """Step a 1-world backend through the case; returns dict of max abs errors.
    Masks / integer fields must match exactly (assert)."""
    a6 = case["agents6"]
    M = a6.shape[0]
    cfg = case["cfg"]
    m_max = int(cfg[3])
    laser = bool(cfg[4])
    obst = case["obstacles"]
    env = make_env(N=1, M=m_max, max_obstacles=max(len(obst), 0), game_over_mode=game_over_mode(cfg),
                   laserscan=laser)
    pad = lambda x, fill=0: np.concatenate([x, np.full((m_max - M,) + x.shape[1:], fill, dtype=x.dtype)])
    a6p = pad(a6)
    a6p[M:, 4] = 1.0
    a6p[M:, 5] = 0.1
    a6p[M:, 0] = 1e3 + np.arange(m_max - M)  # parked far away; inactive anyway
    a6p[M:, 2] = 2e3
    coop = pad(case["coop"], 1.0)[None] if "coop" in case else None  # Agent.cooperation_coef (agent.py:10,103): RVO only
    env.set_scenario(a6p[None], pad(case["policy_id"])[None], pad(case["dynamics_id"])[None],
                     heading0=pad(case["heading0"])[None], n_agents=[M], coop=coop,
                     obstacles=obst[None] if len(obst) else None, n_obst=[len(obst)] if len(obst) else None)
    env.reset()
    T = case["pos"].shape[0] - 1
    ext = case.get("ext_actions")
    errs = {}

    radius = a6[:, 5]

    def close_range_knife(t, tol=1e-9):
        """agents whose nearest-agent gap (index-i-only rule, env.py:649) sits within `tol` of
        GETTING_CLOSE_RANGE = 0.2: the close-penalty branch (env.py:540) is ill-conditioned there."""
        p = case["pos"][t]
        out = np.zeros(M, dtype=bool)
        for i in range(M):
            g = [np.hypot(*(p[i] - p[j])) - radius[i] - radius[j] for j in range(i + 1, M)]
            if g and abs(min(g) - 0.2) < tol:
                out[i] = True
        return out

    def mask_knife(t, tol=1e-9):
        """True when a mask-defining comparison of this step sits within `tol` of its threshold
        (pair collision d <= r_i + r_j, env.py:650; goal d^2 <= 0.75^2, end_conditions.py:4-5;
        timeout, agent.py:187).  The reference's own outcome then depends on last-ulp libm/BLAS
        rounding (e.g. the 6-agent circle preset meets as an exact unit hexagon), so the episode
        is compared only up to the step before."""
        p = case["pos"][t]
        for i in range(M):
            for j in range(i + 1, M):
                if abs(np.hypot(*(p[i] - p[j])) - radius[i] - radius[j]) < tol:
                    return True
        live = ~case["is_done"][t - 1] if t > 0 else np.ones(M, dtype=bool)
        dg = np.hypot(*(p - a6[:, 2:4]).T)
        if (np.abs(dg - 0.75)[live] < tol).any():
            return True
        if (np.abs(case["time_remaining"][t])[live] < tol).any():
            return True
        return False

    def cmp(t):
        for k in (float_keys or FLOAT_KEYS):
            got = np.asarray(env.f(k))[0, :M]
            exp = case[k][t]
            if k == "past_actions":
                exp = exp.reshape(M, 2, 2)
            if k == "reward":
                kn = close_range_knife(t)
                if kn.any():
                    errs["reward_knife_edges"] = errs.get("reward_knife_edges", 0) + int(kn.sum())
                    got = np.where(kn, exp, got)
            diff = got.reshape(exp.shape) - exp
            if k in ANGLE_KEYS:  # +pi and -pi are the same heading: util.wrap's branch there is ulp-dependent
                diff = (diff + np.pi) % (2 * np.pi) - np.pi
            e = np.abs(diff).max() if exp.size else 0.0
            errs[k] = max(errs.get(k, 0.0), float(e))
            tol = reward_tol if (k == "reward" and reward_tol is not None) else ftol
            assert e <= tol * max(1.0, np.abs(exp).max()), (k, t, e, got, exp)
        for k in MASK_KEYS:
            got = np.asarray(env.u(k))[0, :M].astype(bool)
            assert (got == case[k][t]).all(), (k, t, got, case[k][t])
        assert bool(np.asarray(env.u("game_over"))[0]) == bool(case["game_over"][t]), ("game_over", t)
        assert (np.asarray(env.i("step_num"))[0, :M] == case["step_num"][t]).all(), ("step_num", t)
        assert (np.asarray(env.i("num_other_agents_observed"))[0, :M] == case["num_other_agents_observed"][t]).all()
        nobs = case["num_other_agents_observed"][t]
        got, _ = canon_oas(np.asarray(env.f("oas"))[0, :M], nobs, tie)
        exp, nt = canon_oas(case["oas"][t], nobs, tie)
        errs["oas_tie_groups"] = errs.get("oas_tie_groups", 0) + nt
        e = np.abs(got - exp).max()
        errs["oas"] = max(errs.get("oas", 0.0), float(e))
        assert e <= oas_tol * max(1.0, np.abs(exp).max()), ("oas", t, e)
        if laser:
            got = np.asarray(env.f("laserscan"))[0, :M]
            exp = case["laserscan"][t]
            e = np.abs(got - exp).max()
            errs["laserscan"] = max(errs.get("laserscan", 0.0), float(e))
            assert e <= laser_tol, ("laserscan", t, e, got, exp)
        if check:
            check(env, t)

    cmp(0)
    for t in range(T):
        a = None
        if ext is not None:
            a = np.zeros((1, m_max, 2))
            a[0, :M] = ext[t]
        env.step(a)
        if mask_knife(t + 1):
            errs["knife_stop_at"] = t + 1
            break
        cmp(t + 1)
    errs["steps_compared"] = errs.get("knife_stop_at", T + 1)
    return errs


def laser_audit(scan_a, scan_b, pose_a, pose_b, tol=1e-6, base_margin=1e-9):
    """LaserScan is index work (sensors/LaserScanSensor.py:27-58 samples Map.world_coordinates_to_map_indices, Map.py:49-59):
    two backends may only disagree on a beam when the sample that decides it sits on a raster-cell border for the poses they
    hold.  scan_* [N, M, 16]; pose_* = (px, py, heading) arrays [N, M] of the two backends.  Every beam that differs by more than
    `tol` is audited: some sample of that beam (or the ego's own cell, which moves the own-disc mask) must lie within
    `base_margin` + the two backends' pose difference (|dp| + range x |dheading|) of a cell border; a beam that differs without
    such a border is an index error.  Returns (number of differing beams, list of unexplained (world, agent, beam, distance to
    the nearest border in cells, margin in cells))."""
    a, b = np.asarray(scan_a, dtype=np.float64), np.asarray(scan_b, dtype=np.float64)
    bad = np.argwhere(np.abs(a - b) > tol)
    unexplained = []
    rstep, astep = 2 * np.pi / 16, 2 * np.pi / 15
    for n, m, beam in bad:
        px, py, h = (float(np.asarray(v)[n, m]) for v in pose_a)
        qx, qy, qh = (float(np.asarray(v)[n, m]) for v in pose_b)
        ang = (np.pi if beam == 15 else beam * astep - np.pi) + h
        rg = np.arange(16) * rstep
        cells = np.concatenate([10.0 * (px + rg * np.cos(ang)), 10.0 * (py + rg * np.sin(ang)), [10.0 * px, 10.0 * py]])
        dist = float(np.abs(cells - np.rint(cells)).min())
        margin = 10.0 * (base_margin + 2.0 * (abs(px - qx) + abs(py - qy)) + 2.0 * 6.0 * abs((h - qh + np.pi) % (2 * np.pi) - np.pi))
        if dist > margin:
            unexplained.append((int(n), int(m), int(beam), dist, margin))
    return len(bad), unexplained
