#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference env (this container only).

Run:  python tests/golden/make_golden.py   (needs /root/reference; ~1 min)
Output: tests/golden/*.npz  -- pure data (inputs + expected outputs per step).

The reference is imported unmodified through tests/golden/ref_harness.py
(SURVEY.md Appendix C).  Harness conventions, all documented in DESIGN.md:
  * Config.DT = np.float64(0.1) and np.float64 initial headings: the reference was
    written for NumPy 1.x (TensorFlow 1.15 pin, requirements.txt:1) where
    np.float32-scalar (op) python-float promotes to float64.  Under NumPy >= 2
    (NEP 50) the same expressions would stay float32; passing np.float64 operands
    makes NumPy 2.2 reproduce the NumPy-1.x arithmetic the reference intends.
  * D1 (set_agents honoured), D2 (no ig_mcts agent -> no-op), D4 (policy.targetMap=None).
  * ExternalRaw: an ExternalPolicy whose convert_to_action is the identity.
  * LearningD3: LearningPolicy called as (agent, actions[i]).
Every case is stepped past game_over until all agents are done (+ a few steps) so
the done-agent branch of Agent.take_action (agent.py:148-159) is covered.
"""
import os, sys, json
import importlib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_harness as rh

# where the archives are written: tests/golden/ itself, or a scratch directory (tests/test_fixture_repro.py regenerates a group
# there and compares it byte for byte with the committed file)
OUT = os.environ.get("CAGYM_GOLDEN_OUT") or HERE

rh.install_standins()
from gym_collision_avoidance.envs.config import Config  # noqa: E402

Config.ANIMATE_EPISODES = False
Config.DT = np.float64(0.1)
OBS_KEYS = ['dist_to_goal', 'rel_goal', 'radius', 'heading_ego_frame', 'pref_speed', 'other_agents_states']
Config.STATES_IN_OBS = list(OBS_KEYS)

with rh.quiet():
    from gym_collision_avoidance.envs import test_cases as tc
    from gym_collision_avoidance.envs.agent import Agent
    from gym_collision_avoidance.envs.policies.StaticPolicy import StaticPolicy
    from gym_collision_avoidance.envs.policies.NonCooperativePolicy import NonCooperativePolicy
    from gym_collision_avoidance.envs.policies.ExternalPolicy import ExternalPolicy
    from gym_collision_avoidance.envs.policies.LearningPolicy import LearningPolicy
    from gym_collision_avoidance.envs.policies.CARRLPolicy import CARRLPolicy
    from gym_collision_avoidance.envs.dynamics.UnicycleDynamics import UnicycleDynamics
    from gym_collision_avoidance.envs.dynamics.UnicycleDynamicsMaxTurnRate import UnicycleDynamicsMaxTurnRate
    from gym_collision_avoidance.envs.dynamics.UnicycleDynamicsMaxAcc import UnicycleDynamicsMaxAcc
    from gym_collision_avoidance.envs.dynamics.UnicycleSecondOrderEulerDynamics import UnicycleSecondOrderEulerDynamics
    from gym_collision_avoidance.envs.dynamics.FirstOrderDynamics import FirstOrderDynamics
    from gym_collision_avoidance.envs.sensors.OtherAgentsStatesSensor import OtherAgentsStatesSensor
    from gym_collision_avoidance.envs.sensors.LaserScanSensor import LaserScanSensor
    OracleEnv = rh.make_oracle_env_class()

scen = importlib.import_module("gym-exploration-2d_amd.scenarios")


class ExternalRaw(ExternalPolicy):
    def __init__(self):
        ExternalPolicy.__init__(self, str="ExternalRaw")

    def convert_to_action(self, a):
        return a


class LearningD3(LearningPolicy):
    def network_output_to_action(self, idx, agents, actions):
        return LearningPolicy.network_output_to_action(self, agents[idx], actions[idx])


POLICIES = {scen.POLICY_STATIC: StaticPolicy, scen.POLICY_NONCOOP: NonCooperativePolicy,
            scen.POLICY_EXTERNAL: ExternalRaw, scen.POLICY_LEARNING: LearningD3,
            scen.POLICY_CARRL: CARRLPolicy}
DYNAMICS = {scen.DYN_UNICYCLE: UnicycleDynamics, scen.DYN_MAXTURNRATE: UnicycleDynamicsMaxTurnRate,
            scen.DYN_MAXACC: UnicycleDynamicsMaxAcc, scen.DYN_SECONDORDER: UnicycleSecondOrderEulerDynamics,
            scen.DYN_FIRSTORDER: FirstOrderDynamics}


def set_max_agents(m):
    """patch the derived sizes consistently."""
    Config.MAX_NUM_AGENTS_IN_ENVIRONMENT = m
    Config.MAX_NUM_OTHER_AGENTS_IN_ENVIRONMENT = m - 1
    Config.MAX_NUM_OTHER_AGENTS_OBSERVED = m - 1
    Config.STATE_INFO_DICT['other_agents_states']['size'] = (m - 1, 10)


def rect(xl, yl, xu, yu):
    """Obstacle corner order of test_cases.py:2496."""
    return [(xu, yu), (xl, yu), (xl, yl), (xu, yl)]


def snapshot(env, rec, rewards, game_over, laser):
    A = env.agents
    f = lambda name: np.array([float(getattr(a, name)) for a in A])
    v = lambda name: np.array([np.asarray(getattr(a, name), dtype=np.float64) for a in A])
    b = lambda name: np.array([bool(getattr(a, name)) for a in A])
    rec['pos'].append(v('pos_global_frame'))
    rec['vel'].append(v('vel_global_frame'))
    rec['heading'].append(f('heading_global_frame'))
    rec['speed'].append(f('speed_global_frame'))
    rec['delta_heading'].append(f('delta_heading_global_frame'))
    rec['dist_to_goal'].append(f('dist_to_goal'))
    rec['past_dist_to_goal'].append(f('past_dist_to_goal'))
    rec['heading_ego'].append(f('heading_ego_frame'))
    rec['vel_ego'].append(v('vel_ego_frame'))
    rec['ref_prll'].append(v('ref_prll'))
    rec['rel_goal'].append(v('rel_goal'))
    rec['time_remaining'].append(f('time_remaining_to_reach_goal'))
    rec['t'].append(f('t'))
    rec['step_num'].append(np.array([int(a.step_num) for a in A]))
    rec['past_actions'].append(v('past_actions'))
    rec['is_at_goal'].append(b('is_at_goal'))
    rec['was_at_goal_already'].append(b('was_at_goal_already'))
    rec['in_collision'].append(b('in_collision'))
    rec['was_in_collision_already'].append(b('was_in_collision_already'))
    rec['ran_out_of_time'].append(b('ran_out_of_time'))
    rec['is_done'].append(b('is_done'))
    rec['num_other_agents_observed'].append(np.array([int(a.num_other_agents_observed) for a in A]))
    rec['oas'].append(np.array([np.asarray(a.sensor_data['other_agents_states'], dtype=np.float64) for a in A]))
    if laser:
        rec['laserscan'].append(np.array([np.asarray(a.sensor_data['laserscan'], dtype=np.float64) for a in A]))
    rec['reward'].append(np.asarray(rewards, dtype=np.float64) * np.ones(len(A)))
    rec['game_over'].append(bool(game_over))


def _policy_classes():
    """RVO / GA3C policy classes of the reference on top of the delegating stand-ins (rvo2_standin.py, ga3c_standin.py):
    imported lazily, the other fixture groups never touch them."""
    if scen.POLICY_RVO not in POLICIES:
        import rvo2_standin
        import ga3c_standin
        rvo2_standin.install(sys.modules["rvo2"])
        with rh.quiet():
            from gym_collision_avoidance.envs.policies.RVOPolicy import RVOPolicy
            from gym_collision_avoidance.envs.policies.GA3C_CADRL import network
            ga3c_standin.install(network)
            from gym_collision_avoidance.envs.policies.GA3CCADRLPolicy import GA3CCADRLPolicy
        POLICIES[scen.POLICY_RVO] = RVOPolicy
        POLICIES[scen.POLICY_GA3C] = GA3CCADRLPolicy
    return POLICIES


def run_case(agents6, policy_id, dynamics_id, heading0=None, obstacles=(), laser=False,
             ext_actions=None, max_steps=220, extra=3, evaluate=True, homogeneous=False,
             single=False, m_max=10, coop=None):
    agents6 = np.asarray(agents6, dtype=np.float64)
    M = agents6.shape[0]
    policy_id = np.broadcast_to(np.asarray(policy_id), (M,)).astype(np.int32)
    dynamics_id = np.broadcast_to(np.asarray(dynamics_id), (M,)).astype(np.int32)
    if heading0 is None:
        heading0 = scen.heading_toward_goal(agents6)
    heading0 = np.asarray(heading0, dtype=np.float64)
    set_max_agents(m_max)
    Config.EVALUATE_MODE = evaluate
    Config.HOMOGENEOUS_TESTING = homogeneous
    Config.TRAIN_SINGLE_AGENT = False  # rewards for all M agents; scalar mode is rewards[0] (env.py:565-566)
    sensors = [OtherAgentsStatesSensor] + ([LaserScanSensor] if laser else [])
    has_rvo = bool((policy_id == scen.POLICY_RVO).any())
    has_ga3c = bool((policy_id == scen.POLICY_GA3C).any())
    if has_rvo or has_ga3c:
        _policy_classes()
        import rvo2_standin
        import ga3c_standin
    coop = np.ones(M) if coop is None else np.broadcast_to(np.asarray(coop, dtype=np.float64), (M,)).copy()
    with rh.quiet():
        agents = [Agent(agents6[i, 0], agents6[i, 1], agents6[i, 2], agents6[i, 3], agents6[i, 5], agents6[i, 4],
                        np.float64(heading0[i]), POLICIES[int(policy_id[i])], DYNAMICS[int(dynamics_id[i])],
                        sensors, i, cooperation_coef=float(coop[i])) for i in range(M)]
        for a in agents:
            a.policy.targetMap = None
            if int(policy_id[a.id]) == scen.POLICY_GA3C:
                a.policy.initialize_network()  # as test_cases.py:541 / env.py:454 do
        env = OracleEnv()
        env.oracle_obstacles = [rect(*o) for o in obstacles]
        env.set_agents(agents)
        env.reset()
    # game_over rule under test (env.py:722-736); TRAIN_SINGLE_AGENT only affects game_over here
    Config.TRAIN_SINGLE_AGENT = single
    keys = ['pos', 'vel', 'heading', 'speed', 'delta_heading', 'dist_to_goal', 'past_dist_to_goal', 'heading_ego',
            'vel_ego', 'ref_prll', 'rel_goal', 'time_remaining', 't', 'step_num', 'past_actions', 'is_at_goal',
            'was_at_goal_already', 'in_collision', 'was_in_collision_already', 'ran_out_of_time', 'is_done',
            'num_other_agents_observed', 'oas', 'reward', 'game_over'] + (['laserscan'] if laser else [])
    rec = {k: [] for k in keys}
    snapshot(env, rec, np.zeros(M), False, laser)
    T = max_steps if ext_actions is None else min(max_steps, len(ext_actions))
    after = None
    used = []
    # what the reference handed to its RVO simulators / its network at every step (stand-in logs)
    sim = {k: [] for k in ('called', 'pos', 'vel', 'radius', 'pref', 'max_speed', 'collab', 'n_rects', 'n_added', 'new_pos')}
    net = {k: [] for k in ('called', 'x', 'p', 'action')}
    sim_params = None
    for s in range(T):
        if ext_actions is None:
            acts = {}
        else:
            acts = {i: (int(ext_actions[s, i, 0]) if policy_id[i] == scen.POLICY_CARRL else ext_actions[s, i])
                    for i in range(M)}
        # rewards for all agents: the reference slices rewards[0] only when TRAIN_SINGLE_AGENT
        Config.TRAIN_SINGLE_AGENT = False
        with rh.quiet():
            # compute game_over under the requested mode without changing rewards
            orig = env._check_which_agents_done

            def patched():
                Config.TRAIN_SINGLE_AGENT = single
                try:
                    return orig()
                finally:
                    Config.TRAIN_SINGLE_AGENT = False
            env._check_which_agents_done = patched
            if has_rvo:
                del rvo2_standin.LOG[:]
            if has_ga3c:
                del ga3c_standin.LOG[:]
            was_done = [bool(a.is_done) for a in env.agents]
            _, rewards, game_over, info = env.step(acts)
            env._check_which_agents_done = orig
        if has_rvo:
            # one doStep() per live RVO agent, in agent order (env.py:296-320)
            egos = [i for i in range(M) if policy_id[i] == scen.POLICY_RVO and not was_done[i]]
            assert len(rvo2_standin.LOG) == len(egos), (len(rvo2_standin.LOG), egos)
            rec_s = {'called': np.zeros(M, dtype=bool), 'pos': np.zeros((M, M, 2), np.float32),
                     'vel': np.zeros((M, M, 2), np.float32), 'radius': np.zeros((M, M), np.float32),
                     'pref': np.zeros((M, 2), np.float32), 'max_speed': np.zeros(M, np.float32),
                     'collab': np.zeros(M, np.float32), 'n_rects': np.zeros(M, np.int32),
                     'n_added': np.zeros(M, np.int32), 'new_pos': np.zeros((M, 2), np.float32)}
            for i, lg in zip(egos, rvo2_standin.LOG):
                assert lg['collab_set'] == [i]  # only the ego's coefficient is ever set (RVOPolicy.py:85)
                rec_s['called'][i] = True
                rec_s['pos'][i], rec_s['vel'][i], rec_s['radius'][i] = lg['pos'], lg['vel'], lg['radius']
                rec_s['pref'][i], rec_s['max_speed'][i], rec_s['collab'][i] = lg['pref'][i], lg['max_speed'][i], lg['collab'][i]
                rec_s['n_rects'][i], rec_s['n_added'][i] = len(lg['rects']), lg['n_added']
                rec_s['new_pos'][i] = env.agents[i].policy.new_rvo_pos
                if len(lg['rects']):
                    assert np.array_equal(lg['rects'], np.asarray(obstacles, dtype=np.float64).reshape(-1, 4))
                sim_params = lg['params'] if sim_params is None else sim_params
                assert np.array_equal(sim_params, lg['params'])
            for k in sim:
                sim[k].append(rec_s[k])
        if has_ga3c:
            egos = [i for i in range(M) if policy_id[i] == scen.POLICY_GA3C and not was_done[i]]
            assert len(ga3c_standin.LOG) == len(egos)
            rec_n = {'called': np.zeros(M, dtype=bool), 'x': np.zeros((M, 75)), 'p': np.zeros((M, 11)),
                     'action': np.zeros((M, 2))}
            for i, (x, p_) in zip(egos, ga3c_standin.LOG):
                rec_n['called'][i] = True
                rec_n['x'][i], rec_n['p'][i] = x[0], p_[0]
                rec_n['action'][i] = env.agents[i].past_actions[0]  # the fp32-rounded action the env applied
            for k in net:
                net[k].append(rec_n[k])
        snapshot(env, rec, rewards, game_over, laser)
        used.append(s)
        if all(a.is_done for a in env.agents):
            after = extra if after is None else after - 1
            if after == 0:
                break
    out = {k: np.array(v) for k, v in rec.items()}
    out['agents6'] = agents6
    out['heading0'] = heading0
    out['policy_id'] = policy_id
    out['dynamics_id'] = dynamics_id
    out['obstacles'] = np.asarray(obstacles, dtype=np.float64).reshape(-1, 4)
    if ext_actions is not None:
        out['ext_actions'] = np.asarray(ext_actions[:len(used)], dtype=np.float64)
    out['coop'] = coop
    if has_rvo:
        for k, v in sim.items():
            out['sim_' + k] = np.array(v)
        out['sim_params'] = sim_params  # timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst
    if has_ga3c:
        for k, v in net.items():
            out['net_' + k] = np.array(v)
        assert ext_actions is None
        out['ext_actions'] = out['net_action'].copy()  # a backend without the network replays these (POLICY_GA3C = external)
    out['cfg'] = np.array([int(evaluate), int(homogeneous), int(single), int(m_max), int(laser)], dtype=np.int32)
    return out


def save(group, cases):
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[name + "__" + k] = v
    path = os.path.join(OUT, group + ".npz")
    np.savez_compressed(path, **flat)
    print("%-28s %3d cases %8.1f KB" % (group, len(cases), os.path.getsize(path) / 1024))


def f32exact(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def static_mixes():
    """Group B on its own (it draws no random numbers): `--static-mixes-only` regenerates just this archive - what
    tests/test_fixture_repro.py compares byte for byte with the committed file."""
    U, NC, ST = scen.DYN_UNICYCLE, scen.POLICY_NONCOOP, scen.POLICY_STATIC
    cases = {}
    with rh.quiet():
        P4, P6 = tc.preset_testCases(4), tc.preset_testCases(6)
    cases["n4_static_odd"] = run_case(P4[6 if len(P4) > 6 else 0], [NC, ST, NC, ST], U, homogeneous=True)
    cases["n4_static_even"] = run_case(P4[6 if len(P4) > 6 else 0], [ST, NC, ST, NC], U, homogeneous=True)
    cases["n6_static_mix"] = run_case(P6[0], [NC, NC, ST, NC, ST, NC], U, homogeneous=True)
    # a static agent parked on a NonCooperative agent's path: collision with static as member i and j
    blk = np.array([[-3, 0, 3, 0, 1.0, 0.5], [0, 0.2, 5, 5, 1.0, 0.5], [3, 0.1, -3, 0.1, 1.0, 0.4]])
    cases["n3_static_block_j"] = run_case(blk, [NC, ST, NC], U, homogeneous=True)
    cases["n3_static_block_i"] = run_case(blk[[1, 0, 2]], [ST, NC, NC], U, homogeneous=True)
    save("static_mixes", cases)
    return P4, P6, blk


def main():
    rng = np.random.default_rng(20250222)
    U, NC, ST = scen.DYN_UNICYCLE, scen.POLICY_NONCOOP, scen.POLICY_STATIC

    # A. preset_testCases (test_cases.py:2035-2199), NonCooperative / Unicycle
    cases = {}
    for n in (2, 3, 4, 5, 6):
        with rh.quiet():
            P = tc.preset_testCases(n)
        for ci, case in enumerate(P):
            cases["n%d_c%d" % (n, ci)] = run_case(case, NC, U)
    save("presets_noncoop", cases)

    cases = {}
    with rh.quiet():
        P10 = tc.preset_testCases(10)
    for ci, case in enumerate(P10[:2]):
        cases["n10_c%d" % ci] = run_case(case, NC, U, homogeneous=True)
    with rh.quiet():
        P20 = tc.preset_testCases(20)
    cases["n20_c0"] = run_case(P20[0], NC, U, m_max=20, homogeneous=True, max_steps=120)
    save("presets_large", cases)

    # B. Static / NonCooperative mixes: Q8 (static as pair member j), Q9 (timeout reward), timeouts
    P4, P6, blk = static_mixes()

    # A'/B'. the same presets with a seeded perturbation of starts/goals: the exact presets meet at
    # ulp-level knife edges (d == r_i + r_j exactly), where the reference's own masks depend on
    # libm rounding; the perturbed copies exercise collisions/goals/timeouts away from them.
    cases = {}
    pr = np.random.default_rng(31337)

    def jitter(c):
        c = np.array(c, dtype=np.float64)
        c[:, 0:4] += pr.uniform(-0.07, 0.07, c[:, 0:4].shape)
        return c
    for n in (2, 3, 4, 5, 6):
        with rh.quiet():
            P = tc.preset_testCases(n)
        for ci, case in enumerate(P):
            cases["n%d_c%d" % (n, ci)] = run_case(jitter(case), NC, U, homogeneous=True)
    save("presets_perturbed", cases)
    cases = {}
    cases["n4_static_odd"] = run_case(jitter(P4[6 if len(P4) > 6 else 0]), [NC, ST, NC, ST], U, homogeneous=True)
    cases["n4_static_even"] = run_case(jitter(P4[6 if len(P4) > 6 else 0]), [ST, NC, ST, NC], U, homogeneous=True)
    cases["n6_static_mix"] = run_case(jitter(P6[0]), [NC, NC, ST, NC, ST, NC], U, homogeneous=True)
    cases["n3_static_block_j"] = run_case(jitter(blk), [NC, ST, NC], U, homogeneous=True)
    cases["n3_static_block_i"] = run_case(jitter(blk[[1, 0, 2]]), [ST, NC, NC], U, homogeneous=True)
    cases["n10_circle"] = run_case(jitter(P10[0]), NC, U, homogeneous=True)
    save("static_mixes_perturbed", cases)

    # C. dynamics variants with external raw actions and with NonCooperative
    cases = {}
    for dname, d in (("maxturn", scen.DYN_MAXTURNRATE), ("maxacc", scen.DYN_MAXACC),
                     ("second", scen.DYN_SECONDORDER), ("first", scen.DYN_FIRSTORDER), ("uni", scen.DYN_UNICYCLE)):
        w = scen.random_world(np.random.default_rng(77), 4)
        T = 60
        acts = np.stack([rng.uniform(-0.2, 1.2, (T, 4)), rng.uniform(-1.0, 1.0, (T, 4))], axis=-1)
        cases["ext_" + dname] = run_case(w, scen.POLICY_EXTERNAL, d, ext_actions=f32exact(acts), max_steps=T)
        cases["nc_" + dname] = run_case(P4[2], NC, d, max_steps=80, homogeneous=True)
    save("dynamics_variants", cases)

    # D. action maps: Learning (D3), CARRL table, mixes; game_over modes
    cases = {}
    w = scen.random_world(np.random.default_rng(5), 4)
    T = 70
    u = np.stack([rng.uniform(0, 1, (T, 4)), rng.uniform(0.35, 0.65, (T, 4))], axis=-1)
    cases["learning_train"] = run_case(w, scen.POLICY_LEARNING, U, ext_actions=f32exact(u), max_steps=T,
                                       evaluate=False, single=False,
                                       heading0=rng.uniform(-np.pi, np.pi, 4))
    cases["learning_single"] = run_case(w, [scen.POLICY_LEARNING, NC, NC, ST], U, ext_actions=f32exact(u),
                                        max_steps=T, evaluate=False, single=True)
    cases["learning_mixed_train"] = run_case(w, [NC, scen.POLICY_LEARNING, ST, scen.POLICY_LEARNING], U,
                                             ext_actions=f32exact(u), max_steps=T, evaluate=False, single=False)
    d = np.stack([rng.integers(0, 11, (T, 4)).astype(np.float64), np.zeros((T, 4))], axis=-1)
    cases["carrl"] = run_case(w, [scen.POLICY_CARRL, NC, scen.POLICY_CARRL, NC], U, ext_actions=d, max_steps=T)
    save("action_maps", cases)

    # E. random free-space worlds (SURVEY 8(d) rule), NonCooperative, game_over = all done
    cases = {}
    for wi in range(6):
        cases["m4_w%d" % wi] = run_case(scen.random_world(np.random.default_rng(1234 + wi), 4), NC, U,
                                        homogeneous=True)
    for wi in range(3):
        cases["m10_w%d" % wi] = run_case(scen.random_world(np.random.default_rng(1234 + wi), 10), NC, U,
                                         homogeneous=True)
    save("random_worlds", cases)

    # F. obstacles: rasteriser, wall collision, LaserScan (Map.py, LaserScanSensor.py:27-58, env.py:656-666)
    cases = {}
    ig_obst = [(2, 2, 10, 10), (-10, 2, -2, 10), (2, -10, 10, -2), (-10, -10, -2, -2)]  # test_cases.py:3219-3223
    corridor = np.array([[-5, 0, 12, 0, 1.0, 0.5], [0, 0, -12, 0.3, 1.0, 0.5], [5, 0.5, 5, 12, 1.0, 0.5],
                         [0.3, -6, 8, 8, 1.0, 0.4]])
    cases["ig_corridor_nc"] = run_case(corridor, NC, U, obstacles=ig_obst, laser=True, homogeneous=True,
                                       max_steps=150)
    rr = np.random.default_rng(99)
    for wi in range(3):
        obst = []
        for _ in range(rr.integers(2, 6)):
            cx, cy = rr.uniform(-9, 9, 2)
            hw, hh = rr.uniform(0.3, 2.0, 2)
            obst.append((round(cx - hw, 2), round(cy - hh, 2), round(cx + hw, 2), round(cy + hh, 2)))
        w = scen.random_world(np.random.default_rng(4000 + wi), 5)
        cases["rand_rect_w%d" % wi] = run_case(w, NC, U, obstacles=obst, laser=True, homogeneous=True,
                                               max_steps=150)
    # externally driven agents incl. leaving the 30x30 m map (in_map false branch, Map.py:45-47)
    w = np.array([[13.5, 13.0, -5, 0, 1.0, 0.5], [-13.8, 0, 5, 0, 1.0, 0.3], [0, 14.2, 0, -5, 1.0, 0.5]])
    T = 50
    acts = np.stack([np.full((T, 3), 1.0), rng.uniform(-0.3, 0.3, (T, 3))], axis=-1)
    cases["leave_map"] = run_case(w, scen.POLICY_EXTERNAL, U, heading0=[0.6, 3.0, 1.6], obstacles=ig_obst,
                                  laser=True, ext_actions=f32exact(acts), max_steps=T)
    save("obstacles_laserscan", cases)


def ig_primitives():
    """Golden vectors for the information-gain primitives: executed on the reference's
    own Map / edfMap / targetMap / ig_mcts objects, parameters of experiments/src/dmcts.py:74-78."""
    with rh.quiet():
        from gym_collision_avoidance.envs.Map import Map
        from gym_collision_avoidance.envs.information_models.edfMap import edfMap
        from gym_collision_avoidance.envs.policies.ig_mcts import ig_mcts
    rng = np.random.default_rng(4242)
    worlds = {
        "corridor": [(2, 2, 10, 10), (-10, 2, -2, 10), (2, -10, 10, -2), (-10, -10, -2, -2)],  # test_cases.py:3219-3222
        "rects": [(-6.3, 1.2, -3.1, 4.4), (1.7, -8.2, 4.9, -5.5), (5.2, 3.3, 6.1, 9.7), (-1.4, -2.6, 0.8, -1.9)],
    }
    out = {}
    for wname, obst in worlds.items():
        with rh.quiet():
            m = Map(30, 30, 0.1, [rect(*o) for o in obst])

            class Ego(object):
                radius = 0.5
            pol = ig_mcts()
            pol.set_param(ego_agent=Ego(), occ_map=m, map_size=(30, 30), detect_fov=60.0, map_res=0.1,
                          detect_range=5.0, Ntree=5, Nsims=3, parallelize_sims=False, mcts_cp=1., mcts_horizon=4,
                          parallelize_agents=False, dt=0.1, xdt=5, mcts_gamma=0.95, Ncycles=2)
        tm, edf = pol.targetMap, pol.edfMap
        out[wname + "__obstacles"] = np.asarray(obst, dtype=np.float64)
        out[wname + "__edf"] = np.asarray(edf.map, dtype=np.float64)
        # free-space poses (EDF > 0.3), plus the survey's known-answer poses
        poses = [np.array([-5.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.0]), np.array([-5.0, 0.0, 2.5])]
        while len(poses) < 40:
            p = np.append(rng.uniform(-13, 13, 2), rng.uniform(-np.pi, np.pi))
            if edf.get_edf_value_from_pose(p) > 0.3:
                poses.append(p)
        poses = np.array(poses)
        masks = np.zeros((len(poses), 60), dtype=np.uint64)
        for q, p in enumerate(poses):
            for (i, j) in tm.getVisibleCells(p):
                masks[q, j] |= np.uint64(1) << np.uint64(i)
        out[wname + "__vis_poses"] = poses
        out[wname + "__vis_masks"] = masks
        # checkVisibility on random point pairs
        a = rng.uniform(-14, 14, (200, 2))
        b = rng.uniform(-14, 14, (200, 2))
        keep = np.array([edf.get_edf_value_from_pose(x) > 0.05 for x in a])
        a, b = a[keep], b[keep]
        out[wname + "__cv_a"], out[wname + "__cv_b"] = a, b
        out[wname + "__cv_visible"] = np.array([bool(edf.checkVisibility(x, y)) for x, y in zip(a, b)])
        # belief updates: 3 agents, 4 consecutive updates, detections near / far
        bel, obsv, rew = [], [], []
        upd_poses, upd_dets, upd_ndet = [], [], []
        cur = poses[3:6].copy()
        for step in range(4):
            dets = []
            for p in cur:
                d = []
                if rng.uniform() < 0.7:
                    for _ in range(rng.integers(1, 3)):
                        rr, aa = rng.uniform(0.5, 4.5), p[2] + rng.uniform(-0.5, 0.5)
                        d.append(p[0:2] + rr * np.array([np.cos(aa), np.sin(aa)]))
                dets.append(d)
            cells = tm.update([p.copy() for p in cur], [list(d) for d in dets], frame='global')
            mask = np.zeros(60, dtype=np.uint64)
            for (i, j) in cells:
                mask[j] |= np.uint64(1) << np.uint64(i)
            bel.append(tm.map.copy())
            obsv.append(mask)
            rew.append(tm.get_reward_from_cells(cells))
            dd = np.zeros((3, 2, 2))
            nd = np.zeros(3, dtype=np.int32)
            for k, d in enumerate(dets):
                nd[k] = len(d)
                for l, t in enumerate(d):
                    dd[k, l] = t
            upd_poses.append(cur.copy())
            upd_dets.append(dd)
            upd_ndet.append(nd)
            cur = cur + np.array([0.3, 0.1, 0.15])
        out[wname + "__upd_poses"], out[wname + "__upd_dets"] = np.array(upd_poses), np.array(upd_dets)
        out[wname + "__upd_ndet"], out[wname + "__upd_belief"] = np.array(upd_ndet), np.array(bel)
        out[wname + "__upd_observed"], out[wname + "__upd_reward"] = np.array(obsv), np.array(rew)
        # MI of each visibility mask on the final belief
        rws = []
        for q in range(len(poses)):
            cells = {(i, j) for j in range(60) for i in range(60) if (int(masks[q, j]) >> i) & 1}
            rws.append(tm.get_reward_from_cells(cells))
        out[wname + "__mi_reward"] = np.array(rws)
        # get_next_pose for every primitive from every pose
        acts = [np.array([v, w]) for v in (0.0, 2.0, 4.0) for w in (-0.5 * np.pi, 0, 0.5 * np.pi)]
        nxt = np.full((len(poses), 9, 3), np.nan)
        feas = np.zeros((len(poses), 9), dtype=bool)
        for q, p in enumerate(poses):
            for k, a_ in enumerate(acts):
                r = pol.get_next_pose(p.copy(), a_)
                if r is not None:
                    nxt[q, k] = r
                    feas[q, k] = True
        out[wname + "__np_next"], out[wname + "__np_feasible"] = nxt, feas
    path = os.path.join(OUT, "ig_primitives.npz")
    np.savez_compressed(path, **out)
    print("%-28s          %8.1f KB" % ("ig_primitives", os.path.getsize(path) / 1024))


def ga3c_states():
    """State vectors from the reference's own agents_to_ga3c_cadrl_state (policies/GA3CCADRLPolicy.py:45-106),
    called unbound on live Agent objects (TensorFlow is absent, only this numpy half can be executed)."""
    with rh.quiet():
        from gym_collision_avoidance.envs.policies.GA3CCADRLPolicy import GA3CCADRLPolicy
    U, NC = scen.DYN_UNICYCLE, scen.POLICY_NONCOOP
    out = {}

    class Dummy(object):
        pass
    for name, M, seed, steps in (("m4", 4, 11, 25), ("m10", 10, 12, 40), ("m7", 7, 13, 30)):
        w = scen.random_world(np.random.default_rng(seed), M)
        set_max_agents(10)
        Config.EVALUATE_MODE = True
        with rh.quiet():
            agents = [Agent(w[i, 0], w[i, 1], w[i, 2], w[i, 3], w[i, 5], w[i, 4],
                            np.float64(np.arctan2(w[i, 3] - w[i, 1], w[i, 2] - w[i, 0])), POLICIES[NC], DYNAMICS[U],
                            [OtherAgentsStatesSensor], i) for i in range(M)]
            for a in agents:
                a.policy.targetMap = None
            env = OracleEnv()
            env.set_agents(agents)
            env.reset()
        states = []
        for t in range(steps + 1):
            if t:
                with rh.quiet():
                    env.step({})
            st = np.zeros((M, 76))
            for i in range(M):
                st[i] = GA3CCADRLPolicy.agents_to_ga3c_cadrl_state(Dummy(), env.agents[i],
                                                                   env.agents[:i] + env.agents[i + 1:])
            states.append(st)
        out[name + "__agents6"] = w
        out[name + "__states"] = np.array(states)
    path = os.path.join(OUT, "ga3c_states.npz")
    np.savez_compressed(path, **out)
    print("%-28s          %8.1f KB" % ("ga3c_states", os.path.getsize(path) / 1024))


def dmcts_reference(n_seeds=6, n_steps=6):
    """Cumulative team reward of the reference's own Dec-MCTS loop (experiments/src/dmcts.py:50-95) on its
    default scenario IG_agent_crossing with a tiny planning budget, for a few np.random seeds.  The planner uses
    the global np.random stream, so this pins only the statistics (range) the planner must reproduce."""
    from gym_collision_avoidance.envs.collision_avoidance_env import CollisionAvoidanceEnv
    Config.EVALUATE_MODE = True
    Config.HOMOGENEOUS_TESTING = False
    Config.TRAIN_SINGLE_AGENT = False
    set_max_agents(10)
    Config.STATES_IN_OBS = ['radius', 'heading_global_frame', 'pos_global_frame', 'pref_speed', 'other_agents_states']
    out = {"cum_reward": [], "first_actions": [], "pos": []}
    for seed in range(n_seeds):
        np.random.seed(seed)
        with rh.quiet():
            env = CollisionAvoidanceEnv()
            env.reset()
            for i in range(3):
                env.agents[i].policy.set_param(ego_agent=env.agents[i], occ_map=env.map, map_size=(30, 30),
                                               detect_fov=60.0, map_res=0.1, detect_range=5.0, Ntree=5, Nsims=3,
                                               parallelize_sims=False, mcts_cp=1., mcts_horizon=4,
                                               parallelize_agents=False, dt=0.1, xdt=5, mcts_gamma=0.95, Ncycles=2)
        cum, acts, pos = [0.0], [], []
        for t in range(n_steps):
            with rh.quiet():
                env.step({})
            cum.append(cum[-1] + env.agents[0].policy.team_reward)
            acts.append([np.asarray(a.past_actions[0]) for a in env.agents[:3]])
            pos.append([np.append(a.pos_global_frame, a.heading_global_frame) for a in env.agents[:3]])
        out["cum_reward"].append(cum)
        out["first_actions"].append(acts)
        out["pos"].append(pos)
    Config.STATES_IN_OBS = list(OBS_KEYS)
    path = os.path.join(OUT, "ig_dmcts_reference.npz")
    np.savez_compressed(path, **{k: np.array(v, dtype=np.float64) for k, v in out.items()})
    print("%-28s          %8.1f KB" % ("ig_dmcts_reference", os.path.getsize(path) / 1024))
    print(np.array(out["cum_reward"])[:, -1])



def rvo_episodes():
    """UNMODIFIED reference episodes with RVOPolicy agents on top of the delegating rvo2 stand-in (rvo2_standin.py): pins
    RVOPolicy.py:53-117 (simulator inputs incl. the 1.15 radius, pref velocity, collab coefficient; the post-processing to
    (speed, delta heading) with the pi/6 stop-and-turn clamp), one private simulator per RVO agent (Q22), obstacles
    processed once (Q21), done / static / non-cooperative agents as neighbours.  The LP arithmetic itself is the oracle's."""
    U, NC, ST, RVO = scen.DYN_UNICYCLE, scen.POLICY_NONCOOP, scen.POLICY_STATIC, scen.POLICY_RVO
    cases = {}
    with rh.quiet():
        P4, P10 = tc.preset_testCases(4), tc.preset_testCases(10)
    pr = np.random.default_rng(777)

    def jitter(c, amp=0.07):
        c = np.array(c, dtype=np.float64)
        c[:, 0:4] += pr.uniform(-amp, amp, c[:, 0:4].shape)
        return c
    cases["m4_cross"] = run_case(jitter(P4[2]), RVO, U, homogeneous=True, coop=0.5)
    c = jitter(P4[0])
    cases["m4_swap_coop"] = run_case(c, RVO, U, homogeneous=True, coop=[1.0, 0.5, 0.0, 0.3][:len(c)])
    cases["m10_circle"] = run_case(jitter(P10[0]), RVO, U, homogeneous=True, coop=0.5, max_steps=260)
    for wi in range(3):
        w = scen.random_world(np.random.default_rng(9100 + wi), 10)
        cases["m10_w%d" % wi] = run_case(w, RVO, U, homogeneous=True, coop=0.5, max_steps=260)
    w = scen.random_world(np.random.default_rng(9200), 4)
    cases["m4_w0"] = run_case(w, RVO, U, homogeneous=True, coop=0.5)
    # mixes: static and non-cooperative neighbours, agents that finish early stay neighbours (velocity zero)
    w = scen.random_world(np.random.default_rng(9300), 6)
    cases["m6_mix"] = run_case(w, [RVO, NC, ST, RVO, NC, RVO], U, homogeneous=True, coop=[0.5, 1, 1, 0.5, 1, 0.8])
    blk = np.array([[-3, 0, 3, 0, 1.0, 0.5], [0, 0.2, 5, 5, 1.0, 0.5], [3, 0.1, -3, 0.1, 1.0, 0.4], [0.1, -4, 0.2, 4, 1.2, 0.3]])
    cases["m4_static_block"] = run_case(blk, [RVO, ST, RVO, NC], U, homogeneous=True, coop=0.5)
    # a crowd in a small square: linearProgram3 (infeasible LP2), collisions and time-outs among RVO agents
    rr = np.random.default_rng(9400)
    crowd = np.zeros((8, 6))
    crowd[:, 0:2] = rr.uniform(-2.2, 2.2, (8, 2))
    for i in range(8):
        while min([np.hypot(*(crowd[i, 0:2] - crowd[j, 0:2])) for j in range(i)] + [9]) < 1.05:
            crowd[i, 0:2] = rr.uniform(-2.2, 2.2, 2)
    crowd[:, 2:4] = -crowd[:, 0:2] + rr.uniform(-0.5, 0.5, (8, 2))
    crowd[:, 4] = rr.uniform(0.6, 1.4, 8)
    crowd[:, 5] = rr.uniform(0.3, 0.5, 8)
    cases["m8_crowd"] = run_case(crowd, RVO, U, homogeneous=True, coop=0.5, max_steps=260)
    # 20 agents: maxNeighbors = MAX_NUM_AGENTS_IN_ENVIRONMENT = 20 (RVOPolicy.py:15)
    w = scen.random_world(np.random.default_rng(9500), 20)
    cases["m20_w0"] = run_case(w, RVO, U, homogeneous=True, coop=0.5, m_max=20, max_steps=120)
    # rectangles: addObstacle at every call, processObstacles only at the first (RVOPolicy.py:45,56-57; Q21)
    ig_obst = [(2, 2, 10, 10), (-10, 2, -2, 10), (2, -10, 10, -2), (-10, -10, -2, -2)]  # test_cases.py:3219-3223
    corridor = np.array([[-5, 0, 12, 0, 1.0, 0.5], [0, 0, -12, 0.3, 1.0, 0.5], [5, 0.5, 0.4, 12, 1.0, 0.5],
                         [0.3, -6, -8, 0.2, 1.0, 0.4]])
    cases["obst_corridor"] = run_case(corridor, RVO, U, obstacles=ig_obst, laser=True, homogeneous=True, coop=0.5,
                                      max_steps=200)
    ro = np.random.default_rng(9600)
    for wi in range(2):
        obst = []
        for _ in range(ro.integers(2, 5)):
            cx, cy = ro.uniform(-6, 6, 2)
            hw, hh = ro.uniform(0.4, 1.6, 2)
            obst.append((round(cx - hw, 2), round(cy - hh, 2), round(cx + hw, 2), round(cy + hh, 2)))
        w = scen.random_world(np.random.default_rng(9700 + wi), 5)
        # keep starts and goals outside the rectangles (+ clearance)
        for i in range(5):
            for c in (0, 2):
                while any(o[0] - 0.8 < w[i, c] < o[2] + 0.8 and o[1] - 0.8 < w[i, c + 1] < o[3] + 0.8 for o in obst):
                    w[i, c:c + 2] = ro.uniform(-8, 8, 2)
        cases["obst_rand_w%d" % wi] = run_case(w, [RVO, RVO, NC, RVO, RVO], U, obstacles=obst, laser=True,
                                               homogeneous=True, coop=0.5, max_steps=200)
    save("rvo_episodes", cases)


def ga3c_episodes():
    """UNMODIFIED reference episodes with GA3CCADRLPolicy agents whose TensorFlow network is the delegating stand-in
    (ga3c_standin.py -> oracle/ga3c_ref.py): pins GA3CCADRLPolicy.find_next_action (:34-43) - state vector -> predict_p
    -> argmax -> Actions table (network.py:8-17) -> pref_speed scaling - inside an episode.  The network arithmetic is
    the oracle's numpy restatement."""
    U, NC, ST, RVO, GA = scen.DYN_UNICYCLE, scen.POLICY_NONCOOP, scen.POLICY_STATIC, scen.POLICY_RVO, scen.POLICY_GA3C
    cases = {}
    w = scen.random_world(np.random.default_rng(8100), 4)
    cases["m4_all_ga3c"] = run_case(w, GA, U, homogeneous=True, max_steps=200)
    w = scen.random_world(np.random.default_rng(8200), 10)
    cases["m10_ego_ga3c_rvo"] = run_case(w, [GA] + [RVO] * 9, U, homogeneous=False, coop=0.5, max_steps=200)
    w = scen.random_world(np.random.default_rng(8300), 6)
    cases["m6_mix"] = run_case(w, [GA, NC, GA, ST, NC, GA], U, homogeneous=True, max_steps=200)
    with rh.quiet():
        P2 = tc.preset_testCases(2)
    cases["m2_swap"] = run_case(P2[0], GA, U, homogeneous=True, max_steps=200)
    save("ga3c_episodes", cases)


def scenario_statistics(n_worlds=400, n_agents=10):
    """Scenarios drawn by the reference's own train_agents_random_positions (test_cases.py:1362-1463), seeded with
    np.random.seed(k) / random.seed(k) by the function itself.  The third-party rvo2 simulator is absent, so its
    stand-in module gets a constructor-only PyRVOSimulator (RVOPolicy.__init__, RVOPolicy.py:25-28, builds one);
    nothing of it is exercised.  The fixture is data: start/goal rows and the policy class drawn for every agent."""
    import sys as _sys
    from gym_collision_avoidance.envs import test_cases as tc

    class PyRVOSimulator(object):
        def __init__(self, *a, **k):
            pass
    _sys.modules["rvo2"].PyRVOSimulator = PyRVOSimulator
    rows = np.zeros((n_worlds, n_agents, 4))
    noncoop = np.zeros((n_worlds, n_agents), dtype=np.uint8)
    for w in range(n_worlds):
        with rh.quiet():
            agents, _ = tc.train_agents_random_positions(number_of_agents=n_agents, ego_agent_policy=tc.RVOPolicy,
                                                         seed=w + 1)
        assert len(agents) == n_agents
        for i, a in enumerate(agents):
            rows[w, i] = [a.pos_global_frame[0], a.pos_global_frame[1], a.goal_global_frame[0], a.goal_global_frame[1]]
            noncoop[w, i] = type(a.policy).__name__ == "NonCooperativePolicy"
    path = os.path.join(OUT, "scenario_stats.npz")
    np.savez_compressed(path, rows=rows, noncoop=noncoop)
    print("%-28s          %8.1f KB" % ("scenario_stats", os.path.getsize(path) / 1024))


def adapters():
    """Fixtures for the two adapters either side of the hot path, produced by the reference's own
    code: (a) the flat observation MultiagentFlattenDictWrapper.observation (envs/wrappers.py:38-46) makes of the env's
    dict observation, at reset and after every step of one episode (4 agents in a 10-slot env: the absent agents' slots
    are part of the vector); (b) the records add_traj (experiments/src/run_trajectory_dataset_creator.py:53-109) builds
    from the agents' global_state_history of the same episode, flattened to arrays."""
    import importlib.util
    from gym_collision_avoidance.envs.wrappers import MultiagentFlattenDictWrapper
    keys = ['dist_to_goal', 'rel_goal', 'radius', 'heading_ego_frame', 'heading_global_frame', 'pos_global_frame',
            'pref_speed', 'num_other_agents', 'other_agents_states', 'use_ppo']
    old_keys = list(Config.STATES_IN_OBS)
    Config.STATES_IN_OBS = list(keys)
    set_max_agents(10)
    Config.EVALUATE_MODE, Config.HOMOGENEOUS_TESTING, Config.TRAIN_SINGLE_AGENT = True, True, False
    M = 4
    rng = np.random.default_rng(31)
    a6 = scen.random_world(rng, M)
    heading0 = scen.heading_toward_goal(a6)
    pol = np.array([scen.POLICY_NONCOOP, scen.POLICY_NONCOOP, scen.POLICY_STATIC, scen.POLICY_NONCOOP], dtype=np.int32)
    with rh.quiet():
        agents = [Agent(a6[i, 0], a6[i, 1], a6[i, 2], a6[i, 3], a6[i, 5], a6[i, 4], np.float64(heading0[i]),
                        POLICIES[int(pol[i])], UnicycleDynamics, [OtherAgentsStatesSensor], i) for i in range(M)]
        for k, a in enumerate(agents):
            a.policy.targetMap = None
            a.cooperation_coef = 0.25 * (k + 1)
        env = OracleEnv()
        env.set_agents(agents)
        obs = env.reset()
        wrap = MultiagentFlattenDictWrapper(env, dict_keys=keys, max_num_agents=10)
    flat = [np.asarray(wrap.observation(obs), dtype=np.float64)]
    for _ in range(400):
        with rh.quiet():
            obs, rew, go, info = env.step({})
        flat.append(np.asarray(wrap.observation(obs), dtype=np.float64))
        if all(a.is_done for a in env.agents):
            break
    idx = {k: np.array([wrap.observation_indices[a][k] for a in range(10)]) for k in keys}
    # (b) add_traj on the finished episode, exactly as the creator's main() prepares the agents (:164-167)
    spec = importlib.util.spec_from_file_location(
        "creator", os.path.join(rh.REF_ROOT, "gym_collision_avoidance/experiments/src/run_trajectory_dataset_creator.py"))
    src = open(spec.origin).read()
    ns = {"np": np}
    start = src.index("def add_traj(")
    end = src.index("file_dir_template =")
    exec(compile(src[start:end], spec.origin, "exec"), ns)  # the function's own text, executed in place (module import pulls gym/tf/tqdm)
    for a in env.agents:
        a.global_state_history = a.global_state_history[:a.step_num]
    trajs = []
    last = ns["add_traj"](env.agents, trajs, env.dt_nominal, 5.0, None)
    rec = {}
    for i, tr in enumerate(trajs):
        rec["traj%d__time" % i] = np.array([d["time"] for d in tr])
        rec["traj%d__goal" % i] = np.array([d["pedestrian_goal_position"] for d in tr], dtype=np.float64)
        rec["traj%d__coop" % i] = np.array([d["coop_coef"] for d in tr], dtype=np.float64)
        rec["traj%d__pos" % i] = np.array([d["pedestrian_state"]["position"] for d in tr], dtype=np.float64)
        rec["traj%d__vel" % i] = np.array([d["pedestrian_state"]["velocity"] for d in tr], dtype=np.float64)
        rec["traj%d__other_pos" % i] = np.array([d["other_agents_pos"] for d in tr], dtype=np.float64)
        rec["traj%d__other_vel" % i] = np.array([d["other_agents_vel"] for d in tr], dtype=np.float64)
    path = os.path.join(OUT, "adapters.npz")
    np.savez_compressed(path, agents6=a6, heading0=heading0, policy_id=pol, coop=np.array([a.cooperation_coef for a in env.agents]),
                        keys=np.array(keys), flat=np.array(flat), n_traj=np.array(len(trajs)), last_time=np.array(last),
                        step_num=np.array([a.step_num for a in env.agents]),
                        **{"idx__" + k: v for k, v in idx.items()}, **rec)
    Config.STATES_IN_OBS = old_keys
    print("%-28s          %8.1f KB  (%d steps, flat width %d)" % ("adapters", os.path.getsize(path) / 1024, len(flat) - 1, flat[0].size))


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.endswith("-only")]
    if "--dmcts-only" in only:
        dmcts_reference()
    elif "--scenarios-only" in only:
        scenario_statistics()
    elif "--ga3c-only" in only:
        ga3c_states()
    elif "--adapters-only" in only:
        adapters()
    elif "--ig-only" in only:
        ig_primitives()
    elif "--rvo-only" in only:
        rvo_episodes()
    elif "--ga3c-episodes-only" in only:
        ga3c_episodes()
    elif "--static-mixes-only" in only:
        static_mixes()
    else:  # the whole recipe, end to end
        main()
        rvo_episodes()
        ga3c_episodes()
        ig_primitives()
        ga3c_states()
        dmcts_reference()
        scenario_statistics()
        adapters()
