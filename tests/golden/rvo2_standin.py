"""Delegating stand-in for the absent third-party `rvo2` module (THIS CONTAINER ONLY, fixture generation).

The reference's RVOPolicy (policies/RVOPolicy.py) drives a private `rvo2.PyRVOSimulator` through setters, calls
`doStep()` and reads the ego's new position back.  Python-RVO2 is not in the tree and cannot be installed, so its
linear-program arithmetic stays PARITY UNPINNED.  What CAN be pinned is everything the reference's own Python does
around that call.  This stand-in therefore
  * stores what the reference hands over (`addObstacle`, `processObstacles`, `addAgent`, `setAgent*`,
    `setAgentCollabCoeff`), narrowing to C float exactly where the Cython binding of Python-RVO2 narrows,
  * on `doStep()` lets the oracle's restatement of the library (`cao_rvo2_step_agent`, oracle/cagym_oracle.c) move
    every agent of the simulator, and
  * logs the setter state at every `doStep()` so that the fixtures also hold the simulator inputs.
make_golden.py then runs UNMODIFIED reference episodes with RVOPolicy agents on top of it.
Obstacles follow the library's rule: only polygons added before `processObstacles()` are in the obstacle tree
(RVOPolicy.py:45 processes once; the re-additions of every later call never are: SURVEY Q21).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

LOG = []          # one record per doStep(): dict of the simulator state handed over by the reference
DEFAULT_COLLAB = 0.5   # Config.RVO_COLLAB_COEFF (config.py:67); only the ego's coefficient is ever read back


def _f32(x):
    return np.float32(x)


class PyRVOSimulator(object):
    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed,
                 velocity=(0.0, 0.0)):
        self.time_step = _f32(timeStep)
        self.neighbor_dist = _f32(neighborDist)
        self.max_neighbors = int(maxNeighbors)
        self.time_horizon = _f32(timeHorizon)
        self.time_horizon_obst = _f32(timeHorizonObst)
        self.def_radius = _f32(radius)
        self.def_max_speed = _f32(maxSpeed)
        self.def_velocity = (_f32(velocity[0]), _f32(velocity[1]))
        self.pos, self.vel, self.pref, self.radius, self.max_speed, self.collab = [], [], [], [], [], []
        self.pending_obstacles = []      # polygons added since the last processObstacles()
        self.processed_obstacles = []    # rectangles (xl, yl, xu, yu) the obstacle tree holds
        self.n_added_obstacles = 0
        self.collab_set = set()

    # ---- construction -------------------------------------------------------------------------------------------
    def addAgent(self, pos, *args):
        assert not args, "the reference only uses addAgent(pos)"
        self.pos.append([_f32(pos[0]), _f32(pos[1])])
        self.vel.append(list(self.def_velocity))
        self.pref.append([_f32(0.0), _f32(0.0)])
        self.radius.append(self.def_radius)
        self.max_speed.append(self.def_max_speed)
        self.collab.append(_f32(DEFAULT_COLLAB))
        return len(self.pos) - 1

    def addObstacle(self, vertices):
        self.pending_obstacles.append([(float(v[0]), float(v[1])) for v in vertices])
        self.n_added_obstacles += 1
        return self.n_added_obstacles - 1

    def processObstacles(self):
        for poly in self.pending_obstacles:
            # the reference's obstacles are axis-aligned rectangles in the corner order of test_cases.py:2496:
            # [(xu, yu), (xl, yu), (xl, yl), (xu, yl)] (counter-clockwise)
            assert len(poly) == 4
            (xu, yu), (xl, yu2), (xl2, yl), (xu2, yl2) = poly
            assert xu == xu2 and xl == xl2 and yu == yu2 and yl == yl2 and xl < xu and yl < yu, poly
            self.processed_obstacles.append((xl, yl, xu, yu))
        self.pending_obstacles = []

    # ---- setters (Cython narrows every number to float) ------------------------------------------------------------
    def setAgentMaxSpeed(self, i, s):
        self.max_speed[i] = _f32(s)

    def setAgentRadius(self, i, r):
        self.radius[i] = _f32(r)

    def setAgentPosition(self, i, p):
        self.pos[i] = [_f32(p[0]), _f32(p[1])]

    def setAgentVelocity(self, i, v):
        self.vel[i] = [_f32(v[0]), _f32(v[1])]

    def setAgentPrefVelocity(self, i, v):
        self.pref[i] = [_f32(v[0]), _f32(v[1])]

    def setAgentCollabCoeff(self, i, c):
        self.collab[i] = _f32(c)
        self.collab_set.add(i)

    # ---- getters ------------------------------------------------------------------------------------------------
    def getAgentPosition(self, i):
        return (float(self.pos[i][0]), float(self.pos[i][1]))

    def getAgentVelocity(self, i):
        return (float(self.vel[i][0]), float(self.vel[i][1]))

    def getNumAgents(self):
        return len(self.pos)

    # ---- the library call ---------------------------------------------------------------------------------------
    def doStep(self):
        from oracle import oracle as orc
        L = orc.lib()
        n = len(self.pos)
        pos = np.ascontiguousarray(np.array(self.pos, dtype=np.float32).reshape(n, 2))
        vel = np.ascontiguousarray(np.array(self.vel, dtype=np.float32).reshape(n, 2))
        rad = np.ascontiguousarray(np.array(self.radius, dtype=np.float32))
        pref = np.ascontiguousarray(np.array(self.pref, dtype=np.float32).reshape(n, 2))
        rects = np.ascontiguousarray(np.array(self.processed_obstacles, dtype=np.float64).reshape(-1, 4))
        LOG.append(dict(pos=pos.copy(), vel=vel.copy(), radius=rad.copy(), pref=pref.copy(),
                        max_speed=np.array(self.max_speed, dtype=np.float32),
                        collab=np.array(self.collab, dtype=np.float32), collab_set=sorted(self.collab_set),
                        rects=rects.copy(), n_added=self.n_added_obstacles,
                        params=np.array([self.time_step, self.neighbor_dist, self.max_neighbors, self.time_horizon,
                                         self.time_horizon_obst], dtype=np.float64)))
        new_pos = np.zeros((n, 2), dtype=np.float32)
        new_vel = np.zeros((n, 2), dtype=np.float32)
        fp = lambda a: a.ctypes.data_as(C.c_void_p)
        with np.errstate(all="ignore"):
            for a in range(n):  # RVOSimulator::doStep: every agent computes from the OLD state, then all move
                pv = np.ascontiguousarray(pref[a])
                L.cao_rvo2_step_agent(n, a, fp(pos), fp(vel), fp(rad), fp(pv), C.c_float(self.max_speed[a]),
                                      C.c_float(self.collab[a]), C.c_float(self.neighbor_dist), self.max_neighbors,
                                      C.c_float(self.time_horizon), C.c_float(self.time_horizon_obst),
                                      C.c_float(self.time_step), fp(rects) if len(rects) else None, len(rects),
                                      fp(new_pos[a]), fp(new_vel[a]), None, None)
        self.pos = [[new_pos[a, 0], new_pos[a, 1]] for a in range(n)]
        self.vel = [[new_vel[a, 0], new_vel[a, 1]] for a in range(n)]


def install(module):
    """Put the delegating simulator into the (empty) rvo2 stand-in module ref_harness installed."""
    module.PyRVOSimulator = PyRVOSimulator
    from oracle import oracle as orc
    L = orc.lib()
    L.cao_rvo2_step_agent.restype = None
    L.cao_rvo2_step_agent.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float] * 3 + [C.c_int] + [C.c_float] * 3
                                      + [C.c_void_p, C.c_int] + [C.c_void_p] * 4)
