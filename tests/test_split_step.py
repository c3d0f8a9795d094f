"""The split step (cagym_step_begin + cagym_step_finish, csrc/cagym_split3.h) against the fused one-step launch
(cagym_step / cagym_step_autoreset): the RVO half of env.step() - half-planes and linear programs, policies/RVOPolicy.py:53-117 -
runs as its own launch BEFORE the external actions exist (env.py:287-340 gathers every action before any agent moves), the rest
follows.  Same phase functions, same operands, same order: every state field and every output must be BIT-identical, step after
step, with auto-reset, for every kernel specialisation, with and without rectangles / LaserScan, and on a side stream beside a
device policy (step_overlapped).  The fused launch itself is held against the oracle and the reference fixtures elsewhere."""
import importlib

import numpy as np
import pytest

from test_hip_parity import _hip

scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
pytestmark = pytest.mark.gpu


def _pair(N, M, K, laser, pol, seed, n_agents=None, mode=0):
    """two handles on the same scenario pool"""
    S = 3 * N
    envs = []
    if K:
        a6, obst, n_obst, _ = scen.obstacle_worlds(S, M, K, seed=seed)
    else:
        a6, obst, n_obst = scen.random_worlds_fast(S, M, seed=seed), None, None
    for _ in range(2):
        e = _hip(N=N, M=M, max_obstacles=K, game_over_mode=mode, laserscan=laser, n_scenarios=S)
        e.set_scenario(a6, pol(S, M), scen.DYN_UNICYCLE, n_agents=n_agents, coop=np.full((S, M), 0.5), obstacles=obst, n_obst=n_obst)
        e.reset()
        envs.append(e)
    return envs


def _same(a, b, what):
    import torch
    torch.cuda.synchronize()
    sa, sb = a.env.state(), b.env.state()
    for k in sa:
        if k == "map_bits":
            continue
        assert torch.equal(sa[k], sb[k]), (what, "state", k)
    for k in ("obs_oas", "obs_ego", "reward", "flags", "game_over"):
        assert torch.equal(getattr(a.env, k), getattr(b.env, k)), (what, k)
    if a.env.laserscan:
        assert torch.equal(a.env.obs_laser, b.env.obs_laser), (what, "laserscan")


def _mixed(rvo_share):
    def pol(S, M):
        rng = np.random.default_rng(S + M)
        p = np.where(rng.uniform(size=(S, M)) < rvo_share, scen.POLICY_RVO, scen.POLICY_NONCOOP).astype(np.int32)
        p[rng.uniform(size=(S, M)) < 0.05] = scen.POLICY_STATIC
        p[:, 0] = scen.POLICY_EXTERNAL  # agent 0 driven from outside, like cfg4's GA3C ego
        return p
    return pol


@pytest.mark.parametrize("M,K,laser,wpw", [(10, 10, True, None), (10, 10, True, "5"), (10, 0, False, "4"), (10, 0, False, "5"), (4, 6, True, None),
                                           (20, 6, True, None), (20, 0, False, None), (7, 5, True, None), (13, 0, False, None), (32, 0, False, None)])
def test_split_step_equals_fused_step_bitwise(M, K, laser, wpw, monkeypatch):
    import torch
    if wpw:
        monkeypatch.setenv("CAGYM_WPW10", wpw)
    N, T = 41, 200  # 41: the last workgroup is ragged for every worlds-per-workgroup
    rng = np.random.default_rng(M * 31 + K)
    n_agents = rng.integers(max(2, M - 3), M + 1, 3 * N).astype(np.int32)
    fused, split = _pair(N, M, K, laser, _mixed(0.85), seed=500 + M, n_agents=n_agents, mode=0)  # game over: agent 0 done
    _same(fused, split, "reset")
    dev = fused.env.device
    for t in range(T):
        # agent 0 is driven from outside: towards its goal (it arrives, the world restarts) with a random wobble
        ext = torch.zeros((N, M, 2), dtype=torch.float32, device=dev)
        ext[:, 0, 0] = torch.from_numpy(rng.uniform(0.6, 1.0, N).astype(np.float32)).to(dev)
        ext[:, 0, 1] = (-fused.env.state()["heading_ego"][:, 0]).float().clamp(-0.5, 0.5) + torch.from_numpy(rng.uniform(-0.1, 0.1, N).astype(np.float32)).to(dev)
        auto = t % 3 != 2  # both entry points: with and without the restart of finished worlds
        fused.env.step(ext, auto_reset=auto)
        split.env.step_begin()
        split.env.step_finish(ext, auto_reset=auto)
        _same(fused, split, "step %d" % t)
    assert int(fused.env.state()["stat_episodes"].sum()) > 0  # worlds did restart inside the launches
    for e in (fused, split):
        e.env.close()


def test_split_step_overlapped_with_a_device_policy_equals_fused():
    """cfg4's shape in small: agent 0 GA3C-CADRL (cagym_ga3c_act on the current stream) while the RVO half runs on the side stream."""
    import torch
    GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
    N, M, K, T = 64, 10, 10, 100

    def pol(S, M):
        p = np.full((S, M), scen.POLICY_RVO, dtype=np.int32)
        p[:, 0] = scen.POLICY_GA3C
        return p
    fused, split = _pair(N, M, K, True, pol, seed=91, mode=0)
    pf, ps = GA3C(fused.env), GA3C(split.env)
    xf = torch.zeros((N, M, 2), dtype=torch.float32, device=fused.env.device)
    xs = torch.zeros_like(xf)
    for t in range(T):
        pf.act(xf)
        fused.env.step(xf, auto_reset=True)
        split.env.step_overlapped(ps.act, xs, auto_reset=True)
        assert torch.equal(xf, xs), ("external actions", t)
        _same(fused, split, "step %d" % t)
    assert int(fused.env.state()["stat_episodes"].sum()) > 0


def test_split_step_without_rvo_agents_and_call_order():
    import torch
    N, M = 16, 4
    fused, split = _pair(N, M, 0, False, lambda S, M: np.full((S, M), scen.POLICY_NONCOOP, dtype=np.int32), seed=3, mode=1)
    with pytest.raises(RuntimeError, match="without a cagym_step_begin"):
        split.env.step_finish()
    for t in range(30):
        fused.env.step(auto_reset=True)
        split.env.step_begin()  # nothing to solve: no launch
        split.env.step_finish(auto_reset=True)
        _same(fused, split, "step %d" % t)
    split.env.step_begin()
    split.env.reset()  # a reset voids a pending begin
    with pytest.raises(RuntimeError, match="without a cagym_step_begin"):
        split.env.step_finish()
