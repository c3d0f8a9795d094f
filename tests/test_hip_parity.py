"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against (1) the reference-generated golden vectors and (2) the CPU oracle on seeded batches.

Bars (BASELINE.json north_star): collision/done masks and agent ordering bit-exact; positions and
rewards within 1e-5 (the fp32 observation/reward outputs), fp64 state within 1e-9.
"""
import importlib

import numpy as np
import pytest

import golden_util as gu
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

scen = importlib.import_module("gym-exploration-2d_amd.scenarios")


def _hip(**kw):
    from hip_backend import HipBackend
    return HipBackend(**kw)


@pytest.mark.parametrize("group", gu.all_groups())
def test_hip_matches_reference_golden(group):
    from hip_backend import HIP_FLOAT_KEYS
    cases = gu.load_cases(group)
    worst = {}
    # RVO crowds amplify last-ulp libm differences through the fp32 linear programs (see _compare_batch): 1e-7 there
    ftol = 1e-7 if group in ("rvo_episodes", "ga3c_episodes") else 1e-9
    for name, case in cases.items():
        check = None
        if "sim_called" in case:  # the applied fp32 (speed, delta heading) of every RVO agent the reference asked
            def check(env, t, case=case):
                if t == 0:
                    return
                M = case["agents6"].shape[0]
                got = np.asarray(env.f("action"))[0, :M]
                exp = case["past_actions"][t][:, 0, :]
                called = case["sim_called"][t - 1]
                assert np.abs(got - exp)[called].max(initial=0.0) <= 2e-7, ("rvo action", t)
        errs = gu.replay(case, _hip, ftol=ftol, oas_tol=1e-5, laser_tol=1e-6, float_keys=HIP_FLOAT_KEYS, tie=1e-6,
                         reward_tol=1e-6, check=check)
        for k, v in errs.items():
            if isinstance(v, float):
                worst[k] = max(worst.get(k, 0.0), v)
    print(group, {k: "%.1e" % v for k, v in worst.items()})


MASKS = ["is_at_goal", "was_at_goal_already", "in_collision", "was_in_collision_already", "ran_out_of_time", "is_done"]


def _compare_batch(hip, cpu, N, M, t, tie_ok=True, ftol=1e-9):
    """ftol: 1e-9 by default.  RVO crowds amplify the last-ulp differences between the device's and glibc's
    sincos / atan2 through the fp32 linear programs (observed 1.3e-9 after 42 steps of 20 agents): those tests pass
    1e-7, still 100 times inside north_star's 1e-5."""
    for k in ("pos", "vel", "heading", "heading_ego", "dist_to_goal", "time_remaining", "t"):
        a, b = hip.f(k), cpu.f(k)
        d = a - b
        if k in gu.ANGLE_KEYS:
            d = (d + np.pi) % (2 * np.pi) - np.pi
        assert np.nanmax(np.abs(d)) <= ftol and (np.isnan(a) == np.isnan(b)).all(), (k, t, np.nanmax(np.abs(d)))
    for k in MASKS + ["game_over"]:
        assert (hip.u(k) == cpu.u(k)).all(), (k, t)
    assert (hip.i("step_num") == cpu.i("step_num")).all()
    assert np.allclose(hip.f("reward"), cpu.f("reward"), rtol=0, atol=1e-6), ("reward", t)
    nobs = cpu.i("num_other_agents_observed")
    assert (hip.i("num_other_agents_observed") == nobs).all()
    a, b = hip.f("oas"), cpu.f("oas")
    for w in range(N):
        # rows are matched by the agent they describe; a row may sit elsewhere only where its sort key ties (the order
        # among keys that agree to 1e-9 in fp64 is decided by last-ulp rounding; the fp32 rows cannot show such a tie)
        worst, _ = gu.oas_mismatch(a[w], b[w], nobs[w], tol=1e-5, key_tie=1e-5)
        assert worst <= 1e-5, ("oas", t, w, worst)


@pytest.mark.parametrize("M,policy", [(4, scen.POLICY_NONCOOP), (10, scen.POLICY_NONCOOP), (10, scen.POLICY_RVO),
                                      (4, scen.POLICY_RVO), (20, scen.POLICY_RVO), (7, scen.POLICY_RVO)])
def test_hip_matches_oracle_batch(M, policy):
    """Seeded random worlds (SURVEY 8(d) rule), T steps, HIP vs CPU oracle in lock-step.
    RVO = ORCA kernel vs the oracle's restatement ("parity unpinned" w.r.t. the absent rvo2 lib)."""
    N, T = 96, 60
    a6 = scen.random_worlds_fast(N, M, seed=7 + M)
    coop = np.full((N, M), 0.5)
    hip = _hip(N=N, M=M, game_over_mode=1)
    cpu = orc.OracleEnv(N=N, M=M, game_over_mode=1)
    for e in (hip, cpu):
        e.set_scenario(a6, policy, scen.DYN_UNICYCLE, coop=coop)
        e.reset()
    _compare_batch(hip, cpu, N, M, 0)
    for t in range(T):
        hip.step()
        cpu.step()
        if policy == scen.POLICY_RVO:
            # applied fp32 (speed, delta_heading): equal up to fp32 rounding of libm-ulp differences
            # (a straight-moving agent's delta_heading is a ~1e-14 cancellation residue of two atan2 calls)
            assert np.abs(hip.f("action") - cpu.f("action")).max() <= 2e-7, ("action", t)
        _compare_batch(hip, cpu, N, M, t + 1, ftol=1e-7 if policy == scen.POLICY_RVO else 1e-9)


@pytest.mark.parametrize("M,maxnb", [(10, 3), (20, 0), (20, 10), (32, 0)])
def test_rvo_max_neighbors(M, maxnb):
    """RVO maxNeighbors = Config.MAX_NUM_AGENTS_IN_ENVIRONMENT (policies/RVOPolicy.py:15,25): by default every other agent
    of the world is a neighbour (19 half-planes with 20 agents, two per lane of an LP group); an explicit smaller value
    keeps only the nearest ones.  HIP vs the oracle, both configured alike."""
    N, T = 48, 50
    a6 = scen.random_worlds_fast(N, M, seed=101 + M, side=7.5 if M <= 20 else 11.0)
    coop = np.full((N, M), 0.5)
    hip = _hip(N=N, M=M, game_over_mode=1, rvo_max_neighbors=maxnb)
    cpu = orc.OracleEnv(N=N, M=M, game_over_mode=1, rvo_max_neighbors=maxnb)
    for e in (hip, cpu):
        e.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=coop)
        e.reset()
    for t in range(T):
        hip.step()
        cpu.step()
        assert np.abs(hip.f("action") - cpu.f("action")).max() <= 2e-7, ("action", t)
        _compare_batch(hip, cpu, N, M, t + 1, ftol=1e-7)


def test_mixed_policies_dynamics_and_ragged_worlds():
    """Ragged n_agents (including 1-agent, EMPTY and full worlds), every policy/dynamics id, external actions."""
    N, M, T = 64, 10, 40
    rng = np.random.default_rng(3)
    a6 = scen.random_worlds_fast(N, M, seed=11)
    pol = rng.integers(0, 6, (N, M)).astype(np.int32)
    dyn = rng.integers(0, 5, (N, M)).astype(np.int32)
    n_agents = rng.integers(1, M + 1, N).astype(np.int32)
    n_agents[0] = 1
    n_agents[1] = M
    n_agents[2] = 0
    hip = _hip(N=N, M=M, game_over_mode=2)
    cpu = orc.OracleEnv(N=N, M=M, game_over_mode=2)
    for e in (hip, cpu):
        e.set_scenario(a6, pol, dyn, n_agents=n_agents, coop=np.full((N, M), 0.5))
        e.reset()
    for t in range(T):
        ext = np.stack([rng.uniform(0, 1, (N, M)), rng.uniform(0.3, 0.7, (N, M))], -1).astype(np.float32)
        ext[pol == scen.POLICY_CARRL, 0] = rng.integers(0, 11, (pol == scen.POLICY_CARRL).sum())
        hip.step(ext)
        cpu.step(ext.astype(np.float64))
        assert np.abs(hip.f("action") - cpu.f("action")).max() <= 2e-7, ("action", t)
        _compare_batch(hip, cpu, N, M, t + 1)


def test_rollout_equals_steps_and_autoreset():
    """cagym_rollout (state in registers, auto-reset in kernel) == step() + reset(advance) on the host."""
    N, M, T = 48, 10, 320
    a6 = scen.random_worlds_fast(4 * N, M, seed=5)
    envs = [_hip(N=N, M=M, game_over_mode=1, n_scenarios=4 * N) for _ in range(2)]
    for e in envs:
        e.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((4 * N, M), 0.5))
        e.reset()
    a, b = envs[0].env, envs[1].env
    traj = a.rollout(T, auto_reset=True)
    import torch
    for t in range(T):
        obs, rew, go, info = b.step()
        assert torch.equal(traj["reward"][t], rew), t
        assert torch.equal(traj["flags"][t], info["flags"]), t
        assert torch.equal(traj["game_over"][t], go), t
        if go.any():
            b.reset(world_mask=go, advance_episode=True)
        assert torch.equal(traj["other_agents_states"][t], b.obs_oas), t
        assert torch.equal(traj["ego"][t], b.obs_ego), t
    sa, sb = a.episode_stats(), b.episode_stats()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(sa["stat_episodes"].sum()) > 0
    for k in ("pos_x", "pos_y", "heading", "time_remaining", "status", "episode"):
        assert torch.equal(a.state()[k], b.state()[k]), k


def test_full_size_properties():
    """BASELINE config 4096 x 10 (RVO + OAS): size-independent properties."""
    import torch
    N, M, T = 4096, 10, 64
    a6 = scen.random_worlds_fast(2 * N, M, seed=1234)
    B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    outs = []
    for rep in range(2):
        env = B(N, M, n_scenarios=2 * N, game_over_mode="all")
        env.set_scenarios(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((2 * N, M), 0.5))
        env.reset()
        env.rollout(256, auto_reset=True, out=env.alloc_rollout(256, obs=False))  # let episodes finish
        traj = env.rollout(T, auto_reset=True)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in traj.items()})
        st = {k: v.clone() for k, v in env.state().items()}
        env.close()
    for k in outs[0]:  # determinism: same seed twice -> identical bytes
        assert torch.equal(outs[0][k], outs[1][k]), k
    tr = outs[0]
    oas = tr["other_agents_states"]
    key = oas[..., 8]
    nobs = tr["ego"][..., 9].long()
    # sortedness: rows are farthest -> closest among observed rows
    K = M - 1
    idx = torch.arange(K, device=key.device)
    valid = idx.view(1, 1, 1, K) < nobs.unsqueeze(-1)
    d = key[..., :-1] - key[..., 1:]
    assert bool(((d >= 0) | ~valid[..., 1:]).all())
    assert bool((oas[~valid.unsqueeze(-1).expand_as(oas)] == 0).all())
    # combined radius column and type column
    assert bool(((oas[..., 7] == 1.0) | ~valid).all()) and bool(((oas[..., 9] == 2.0) | ~valid).all())
    # rewards are in the reference's scaled range, flags consistent with DONE
    r = tr["reward"]
    assert float(r.min()) >= -10 / 13 - 1e-6 and float(r.max()) <= 3 / 13 + 1e-6
    fl = tr["flags"]
    done = (fl & 8) != 0
    assert bool((done == ((fl & 7) != 0)).all())
    # ORCA with collab 0.5 in free space: collisions are rare, most agents reach the goal
    ep = int(st["stat_episodes"].sum())
    out = st["stat_outcomes"].sum(0)
    assert ep > 0 and int(out[0]) > 10 * int(out[1])


@pytest.mark.parametrize("M,wpw", [(2, None), (4, None), (7, None), (10, "4"), (10, "5"), (13, None), (20, None), (32, None)])
def test_kernel_generations_agree_bitwise(M, wpw, monkeypatch):
    """The software-pipelined kernels (generation 3, default) and the one-lane-per-agent kernels (generation 1,
    CAGYM_KERNEL=v1) share their arithmetic: trajectories, observations and statistics must be identical.  (Generation 2 was
    retired in round 2; an unknown CAGYM_KERNEL value is an error, see test_unknown_kernel_generation_is_rejected.)"""
    import torch
    N, T = 50, 200
    a6 = scen.random_worlds_fast(3 * N, M, seed=77 + M)
    rng = np.random.default_rng(M)
    pol = np.where(rng.uniform(size=(3 * N, M)) < 0.8, scen.POLICY_RVO, scen.POLICY_NONCOOP).astype(np.int32)
    pol[rng.uniform(size=(3 * N, M)) < 0.05] = scen.POLICY_STATIC
    n_agents = rng.integers(max(1, M - 3), M + 1, 3 * N).astype(np.int32)
    res = []
    if wpw:
        monkeypatch.setenv("CAGYM_WPW10", wpw)  # both worlds-per-workgroup variants of the M = 10 kernels
    for gen in ("v1", "v3"):
        monkeypatch.setenv("CAGYM_KERNEL", gen)
        e = _hip(N=N, M=M, game_over_mode=1, n_scenarios=3 * N)
        e.set_scenario(a6, pol, scen.DYN_UNICYCLE, n_agents=n_agents, coop=np.full((3 * N, M), 0.5))
        e.reset()
        tr = e.env.rollout(T, auto_reset=True)
        obs, rew, go, info = e.env.step()
        torch.cuda.synchronize()
        st = {k: v.clone() for k, v in e.env.state().items()}
        res.append(({k: v.clone() for k, v in tr.items()}, st, e.env.obs_oas.clone(), rew.clone()))
        e.env.close()
    (t1, s1, o1, r1) = res[0]
    for gen, (t2, s2, o2, r2) in zip(("v3",), res[1:]):
        for k in t1:
            assert torch.equal(t1[k], t2[k]), (gen, k)
        for k in s1:
            assert torch.equal(s1[k], s2[k]), (gen, k)
        assert torch.equal(o1, o2) and torch.equal(r1, r2), gen
    assert int(s1["stat_episodes"].sum()) > 0 or M > 20  # 32 crowded agents need more than T steps to all finish


def test_packed_episode_stats_kernel_matches_column_packing():
    """cagym_pack_episode_stats (one kernel, the record the multi-GPU all-gather carries) == stats.pack_episode_stats."""
    import torch
    stats = importlib.import_module("gym-exploration-2d_amd.stats")
    N, M = 37, 10
    a6 = scen.random_worlds_fast(3 * N, M, seed=2)
    e = _hip(N=N, M=M, game_over_mode=1, n_scenarios=3 * N)
    e.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((3 * N, M), 0.5))
    e.reset()
    e.env.rollout(400, auto_reset=True)
    a = e.env.packed_episode_stats()
    b = stats.pack_episode_stats(e.env.episode_stats())
    torch.cuda.synchronize()
    assert a.dtype == torch.int32 and tuple(a.shape) == (N, 6) and torch.equal(a, b)
    assert int(a[:, 1].sum()) > 0


def test_unknown_kernel_generation_is_rejected(monkeypatch):
    """A removed or misspelt kernel generation must not silently run the default one."""
    monkeypatch.setenv("CAGYM_KERNEL", "v2")
    with pytest.raises(Exception):
        _hip(N=4, M=4, game_over_mode=1)


def test_ga3c_device_policy_replays_reference_episodes():
    """GA3CCADRLPolicy.find_next_action (policies/GA3CCADRLPolicy.py:34-43) on the device - selection, state vectors, the
    MFMA forward, argmax, action table, pref_speed scaling (cagym_ga3c_act) - driving the HIP env through the reference-run
    GA3C episodes (tests/golden/ga3c_episodes.npz; the reference's network was the oracle's numpy restatement, so this pins
    the plumbing and the fp32 network against the fp64 one, not TensorFlow).  The chosen action must be the reference's at
    every step unless the reference's own two best probabilities are closer than 2e-4 (fp32 vs fp64 network)."""
    import torch
    from hip_backend import HipBackend
    GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
    n_cmp = 0
    for name, case in gu.load_cases("ga3c_episodes").items():
        a6 = case["agents6"]
        M, m_max = a6.shape[0], int(case["cfg"][3])
        hip = HipBackend(N=1, M=m_max, game_over_mode=gu.game_over_mode(case["cfg"]))
        pad = lambda x, fill=0: np.concatenate([x, np.full((m_max - M,) + x.shape[1:], fill, dtype=x.dtype)])
        a6p = pad(a6)
        a6p[M:, 4], a6p[M:, 5], a6p[M:, 0], a6p[M:, 2] = 1.0, 0.1, 1e3 + np.arange(m_max - M), 2e3
        hip.set_scenario(a6p[None], pad(case["policy_id"])[None], pad(case["dynamics_id"])[None],
                         heading0=pad(case["heading0"])[None], n_agents=[M], coop=pad(case["coop"], 1.0)[None])
        hip.reset()
        pol = GA3C(hip.env, max_observed=m_max - 1)
        for t in range(case["net_called"].shape[0]):
            ext = pol.act()
            torch.cuda.synchronize()
            got = ext[0, :M].double().cpu().numpy()
            ok = True
            for i in np.nonzero(case["net_called"][t])[0]:
                p = np.sort(case["net_p"][t, i])
                if p[-1] - p[-2] < 2e-4:
                    ok = ok and np.abs(got[i] - case["net_action"][t, i]).max() <= 2e-7
                    continue
                assert np.abs(got[i] - case["net_action"][t, i]).max() <= 2e-7, (name, t, i, got[i], case["net_action"][t, i])
                n_cmp += 1
            if not ok:
                break  # a near-tie went the other way: the trajectories part here, legitimately
            hip.env.step(ext)
            for k in ("pos", "heading"):
                d = np.asarray(hip.f(k))[0, :M] - case[k][t + 1]
                if k == "heading":
                    d = (d + np.pi) % (2 * np.pi) - np.pi
                assert np.abs(d).max() <= 1e-7, (name, k, t)
            for k in MASKS:
                assert (np.asarray(hip.u(k))[0, :M].astype(bool) == case[k][t + 1]).all(), (name, k, t)
        hip.env.close()
    assert n_cmp > 700


def test_step_autoreset_equals_rollout_and_graph_replay():
    """cagym_step_autoreset (one launch per step, reset inside) == cagym_rollout, eagerly and replayed from a
    captured HIP graph (no host sync or allocation inside the step entry points)."""
    import torch
    N, M, T = 40, 10, 260
    a6 = scen.random_worlds_fast(3 * N, M, seed=15)
    mk = lambda: _hip(N=N, M=M, game_over_mode=1, n_scenarios=3 * N)
    envs = [mk(), mk(), mk()]
    for e in envs:
        e.set_scenario(a6, scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((3 * N, M), 0.5))
        e.reset()
    a, b, c = (e.env for e in envs)
    traj = a.rollout(T, auto_reset=True)
    # eager single launches
    for t in range(T):
        obs, rew, go, info = b.step(auto_reset=True)
        assert torch.equal(traj["reward"][t], rew) and torch.equal(traj["game_over"][t], go), t
        assert torch.equal(traj["other_agents_states"][t], b.obs_oas) and torch.equal(traj["ego"][t], b.obs_ego), t
    # graph replay
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        c.step(auto_reset=True)  # warm-up launch on the side stream (step 0)
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(traj["reward"][0], c.reward)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c.step(auto_reset=True)
    # the capture itself does not execute the step
    for t in range(1, T):
        g.replay()
        assert torch.equal(traj["reward"][t], c.reward), t
        assert torch.equal(traj["other_agents_states"][t], c.obs_oas), t
    torch.cuda.synchronize()
    for k in ("stat_episodes", "stat_steps", "stat_outcomes", "stat_return"):
        assert torch.equal(a.episode_stats()[k], b.episode_stats()[k]) and torch.equal(a.episode_stats()[k], c.episode_stats()[k]), k
    assert int(a.episode_stats()["stat_episodes"].sum()) > 0
