"""ctypes binding of libcagym_hip.so (include/cagym.h).  No fallback: a missing library raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAGYM_LIB") or os.path.join(HERE, "csrc", "libcagym_hip.so")  # CAGYM_LIB: diagnostic builds

EGO_WIDTH = 12
FLAG_AT_GOAL, FLAG_IN_COLLISION, FLAG_RAN_OUT_OF_TIME, FLAG_DONE = 1, 2, 4, 8
FLAG_WAS_AT_GOAL, FLAG_WAS_IN_COLLISION, FLAG_ACTIVE = 16, 32, 64
GO_AGENT0, GO_ALL, GO_LEARNING = 0, 1, 2


class CagymConfig(C.Structure):
    _fields_ = [("n_worlds", C.c_int32), ("max_agents", C.c_int32), ("n_scenarios", C.c_int32),
                ("max_obstacles", C.c_int32), ("game_over_mode", C.c_int32), ("collide_with_static", C.c_int32),
                ("laserscan", C.c_int32), ("device", C.c_int32), ("dt", C.c_double),
                ("rvo_max_neighbors", C.c_int32), ("reserved", C.c_int32)]


class CagymOutputs(C.Structure):
    _fields_ = [("obs_oas", C.c_void_p), ("obs_ego", C.c_void_p), ("laserscan", C.c_void_p),
                ("reward", C.c_void_p), ("flags", C.c_void_p), ("game_over", C.c_void_p)]


STATE_FIELDS = [("pos_x", "f8"), ("pos_y", "f8"), ("vel_x", "f8"), ("vel_y", "f8"), ("heading", "f8"),
                ("heading_ego", "f8"), ("dist_to_goal", "f8"), ("time_remaining", "f8"), ("t", "f8"),
                ("goal_x", "f8"), ("goal_y", "f8"), ("radius", "f8"), ("pref_speed", "f8"), ("speed", "f8"),
                ("delta_heading", "f8"), ("aux0", "f8"), ("aux1", "f8"), ("action", "f4"), ("status", "u4"),
                ("step_num", "i4"), ("n_agents", "i4"), ("n_observed", "i4"), ("episode", "i4"),
                ("map_bits", "u4"), ("stat_return", "f4"), ("stat_episodes", "i4"), ("stat_steps", "i4"),
                ("stat_outcomes", "i4")]


class CagymStatePtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in STATE_FIELDS]


class CagymGenParams(C.Structure):
    """cagym_gen_params (include/cagym.h)."""
    _fields_ = [("seed", C.c_uint64), ("n_min", C.c_int32), ("n_max", C.c_int32), ("ego_policy", C.c_int32),
                ("ego_dynamics", C.c_int32), ("policy_a", C.c_int32), ("policy_b", C.c_int32),
                ("other_dynamics", C.c_int32), ("max_tries", C.c_int32), ("p_b", C.c_double), ("side", C.c_double),
                ("min_travel", C.c_double), ("min_sep", C.c_double), ("radius", C.c_double), ("pref_speed", C.c_double),
                ("coop", C.c_double)]


class CagymScenarioPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("agents6", "policy", "dynamics", "n_agents", "coop")]


EXPORTS = ["cagym_version", "cagym_create", "cagym_destroy", "cagym_last_error", "cagym_set_scenarios",
           "cagym_reset", "cagym_step", "cagym_step_autoreset", "cagym_step_begin", "cagym_step_finish", "cagym_rollout", "cagym_get_state", "cagym_laserscan",
           "cagym_generate_scenarios", "cagym_get_scenarios", "cagym_occupancy_grid", "cagym_kernel_name"]

_lib = None


def load():
    """Load the HIP library.  torch is imported first so that the library binds to the HIP runtime
    torch already mapped (same SONAME); there is deliberately no CPU / pure-Python fallback."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps torch/lib/libamdhip64.so before our DT_NEEDED is resolved)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libcagym_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python gym-exploration-2d_amd/build.py`; there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.cagym_version.restype = C.c_int
    L.cagym_last_error.restype = C.c_char_p
    L.cagym_last_error.argtypes = [C.c_void_p]
    L.cagym_create.argtypes = [C.POINTER(CagymConfig), C.POINTER(C.c_void_p)]
    L.cagym_destroy.argtypes = [C.c_void_p]
    L.cagym_set_scenarios.argtypes = [C.c_void_p] * 10
    L.cagym_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(CagymOutputs), C.c_void_p]
    L.cagym_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CagymOutputs), C.c_void_p]
    L.cagym_step_autoreset.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CagymOutputs), C.c_void_p]
    L.cagym_rollout.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CagymOutputs), C.c_void_p]
    L.cagym_step_begin.argtypes = [C.c_void_p, C.c_void_p]
    L.cagym_step_finish.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CagymOutputs), C.c_int, C.c_void_p]
    L.cagym_get_state.argtypes = [C.c_void_p, C.POINTER(CagymStatePtrs)]
    L.cagym_laserscan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.cagym_pack_episode_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.cagym_occupancy_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.cagym_generate_scenarios.argtypes = [C.c_void_p, C.POINTER(CagymGenParams), C.POINTER(C.c_int32), C.c_void_p]
    L.cagym_get_scenarios.argtypes = [C.c_void_p, C.POINTER(CagymScenarioPtrs)]
    L.cagym_kernel_name.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    _lib = L
    return L


def check(L, env, rc, what):
    if rc != 0:
        msg = L.cagym_last_error(env)
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
