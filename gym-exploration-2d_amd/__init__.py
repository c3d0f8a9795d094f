"""gym-exploration-2d_amd ("cagym"): MI355X-native batched CollisionAvoidanceEnv.

The env.step() hot path of mlodel/gym-exploration-2d (a fork of mit-acl/gym-collision-avoidance)
as hand-written HIP kernels for gfx950 behind a C ABI (include/cagym.h), with the reference's
gym.Env surface on top.  Importing the package does not load the HIP library; constructing an
env does, and fails loudly if it is missing (no CPU fallback).
"""
from . import scenarios  # noqa: F401
from .scenarios import *  # noqa: F401,F403

__all__ = ["BatchedCollisionAvoidanceEnv", "CollisionAvoidanceEnv", "Config", "scenarios"]


def __getattr__(name):
    if name == "BatchedCollisionAvoidanceEnv":
        from .batched_env import BatchedCollisionAvoidanceEnv
        return BatchedCollisionAvoidanceEnv
    if name in ("CollisionAvoidanceEnv", "Agent", "Config"):
        from . import env as _env
        return getattr(_env, name)
    raise AttributeError(name)
