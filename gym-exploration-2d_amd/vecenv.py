"""Flat-observation / VecEnv adapter on device tensors.

Byte layout of the reference's MultiagentFlattenDictWrapper (envs/wrappers.py:8-46): for agent 0..M-1, for
key in dict_keys (Config.STATES_IN_OBS order): the raveled value -- one float32 row of M * obs_len per world.
VecEnv convention of stable-baselines' DummyVecEnv as the reference uses it (experiments/src/env_utils.py:29-62):
step(actions) -> (obs[n_envs, D], rews[n_envs], dones[n_envs], infos), finished envs restart and return the
first observation of the new episode.  Everything stays on the GPU; no host round trip per step.
"""
import numpy as np
import torch

# (source tensor, first column, width) of every in-scope key (config.py:104-215)
_EGO = {"dist_to_goal": (0, 1), "rel_goal": (1, 2), "radius": (3, 1), "heading_ego_frame": (4, 1),
        "heading_global_frame": (5, 1), "pos_global_frame": (6, 2), "pref_speed": (8, 1), "num_other_agents": (9, 1),
        "use_ppo": (10, 1)}


def key_width(key, M):
    if key in _EGO:
        return _EGO[key][1]
    if key == "other_agents_states":
        return (M - 1) * 10
    if key == "other_agent_states":
        return 10
    if key == "laserscan":
        return 16
    raise KeyError("observation key %r is outside the hot-path scope (SURVEY.md 8(a))" % key)


def observation_indices(dict_keys, M):
    """Same bookkeeping as MultiagentFlattenDictWrapper.__init__ (wrappers.py:17-31)."""
    idx, size = {}, 0
    for agent in range(M):
        idx[agent] = {}
        lo = size
        for key in dict_keys:
            w = key_width(key, M)
            idx[agent][key] = [size, size + w]
            size += w
        idx[agent]["BOUNDS"] = [lo, size]
    return idx, size


class FlatObservation(object):
    def __init__(self, benv, dict_keys):
        self.b, self.keys = benv, list(dict_keys)
        self.indices, self.size = observation_indices(self.keys, benv.M)
        self.agent_size = self.size // benv.M

    def __call__(self, obs=None):
        b = self.b
        parts = []
        for key in self.keys:
            if key in _EGO:
                c, w = _EGO[key]
                parts.append(b.obs_ego[:, :, c:c + w])
            elif key == "other_agents_states":
                parts.append(b.obs_oas.reshape(b.N, b.M, -1))
            elif key == "other_agent_states":
                parts.append(b.obs_oas[:, :, 0, :])
            elif key == "laserscan":
                parts.append(b.obs_laser)
        return torch.cat(parts, dim=2).reshape(b.N, self.size)

    def array_to_dict(self, row):
        """observationArrayToDict (wrappers.py:48-57) for one world's flat row."""
        row = np.asarray(row)
        return {a: {k: row[self.indices[a][k][0]:self.indices[a][k][1]] for k in self.keys} for a in range(self.b.M)}


class CagymVecEnv(object):
    """n_envs = N worlds.  rews: agent 0's reward when single_agent (Config.TRAIN_SINGLE_AGENT, env.py:565-566)
    else [N, M]."""

    def __init__(self, benv, dict_keys, single_agent=True):
        self.b = benv
        self.num_envs = benv.N
        self.flat = FlatObservation(benv, dict_keys)
        self.single_agent = single_agent
        self._actions = None

    def reset(self):
        self.b.reset()
        return self.flat()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self):
        b = self.b
        _, rew, go, info = b.step(self._actions, auto_reset=True)  # DummyVecEnv auto-reset inside the launch
        rews = rew[:, 0] if self.single_agent else rew
        return self.flat(), rews, go.bool(), {"flags": info["flags"]}

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.b.close()
