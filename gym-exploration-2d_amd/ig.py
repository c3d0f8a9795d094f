"""Information-gain planner primitives on device (cagym_ig_* of include/cagym.h).

Counterparts of the reference's targetMap / edfMap objects and of the Dec-MCTS roll-out primitives
(information_models/targetMap.py, information_models/edfMap.py, policies/ig_mcts.py:117-253,
policies/pydecmcts/DecMCTS.py:233-271).  Visibility sets are [.., 60] int64 tensors (bit i of word j
<=> belief cell (i, j)); the tree bookkeeping (UCT select / expand / back-propagate) stays with the caller.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

FOV_DEG60 = 60.0 * np.pi / 180  # detect_fov=60.0 -> targetMap.sensFOV (ig_mcts.py:67)
PRIMITIVES = np.array([[v, w] for v in (0.0, 2.0, 4.0) for w in (-0.5 * np.pi, 0.0, 0.5 * np.pi)])  # ig_mcts.py:247-253


class InfoGain(object):
    def __init__(self, benv, fov_rad=FOV_DEG60, sens_range=5.0, xdt=5, dt=0.1):
        self.b = benv
        self.L = benv.L
        self.fov, self.range, self.xdt, self.dt = float(fov_rad), float(sens_range), int(xdt), float(dt)
        L = self.L
        vp = C.c_void_p
        L.cagym_ig_init.argtypes = [vp, vp]
        L.cagym_ig_reset_belief.argtypes = [vp, vp, vp]
        L.cagym_ig_get.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.cagym_ig_visible_cells.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.c_double, vp, vp]
        L.cagym_ig_update_belief.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp]
        L.cagym_ig_mi_reward.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        L.cagym_ig_next_pose.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp]
        L.cagym_ig_rollouts.argtypes = [vp] * 7 + [C.c_int] * 4 + [C.c_double] * 3 + [C.c_uint64, vp, vp, vp, vp, vp]
        with torch.cuda.device(benv.device):
            _lib.check(L, benv.h, L.cagym_ig_init(benv.h, benv._stream()), "cagym_ig_init")
        d2, bel = vp(), vp()
        _lib.check(L, benv.h, L.cagym_ig_get(benv.h, C.byref(d2), C.byref(bel)), "cagym_ig_get")
        from .batched_env import _DevArray
        self.edf_d2 = torch.as_tensor(_DevArray(d2.value, (benv.S, 300, 300), "i4"), device=benv.device)
        self.belief = torch.as_tensor(_DevArray(bel.value, (benv.N, 60, 60), "f8"), device=benv.device)

    def _t(self, x, dtype, shape=None):
        t = torch.as_tensor(x, device=self.b.device).to(dtype)
        if shape is not None:
            t = t.reshape(shape)
        return t.contiguous()

    def edf(self):
        """[S,300,300] f64 Euclidean distance field in metres (edfMap.map)."""
        return self.edf_d2.double().sqrt() * 0.1

    def reset_belief(self, world_mask=None):
        m = None if world_mask is None else self._t(world_mask, torch.uint8)
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ig_reset_belief(self.b.h, None if m is None else m.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ig_reset_belief")

    def visible_cells(self, poses, world):
        poses = self._t(poses, torch.float64, (-1, 3))
        world = self._t(world, torch.int32, (-1,))
        Q = poses.shape[0]
        masks = torch.empty((Q, 60), dtype=torch.int64, device=self.b.device)
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ig_visible_cells(self.b.h, poses.data_ptr(), world.data_ptr(), Q, self.fov, self.range,
                                               masks.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ig_visible_cells")
        return masks

    def update_belief(self, poses, detections, n_det, n_poses=None):
        """poses [N,P,3]; detections [N,P,Dmax,2] (global positions); n_det [N,P]. Returns observed [N,60]."""
        N = self.b.N
        poses = self._t(poses, torch.float64)
        P = poses.shape[1]
        det = self._t(detections, torch.float64)
        Dmax = det.shape[2]
        nd = self._t(n_det, torch.int32, (N, P))
        npz = None if n_poses is None else self._t(n_poses, torch.int32, (N,))
        obs = torch.empty((N, 60), dtype=torch.int64, device=self.b.device)
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ig_update_belief(self.b.h, poses.data_ptr(), None if npz is None else npz.data_ptr(),
                                               det.data_ptr(), nd.data_ptr(), P, Dmax, self.fov, self.range,
                                               obs.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ig_update_belief")
        return obs

    def mi_reward(self, masks, world):
        masks = self._t(masks, torch.int64, (-1, 60))
        world = self._t(world, torch.int32, (-1,))
        out = torch.empty((masks.shape[0],), dtype=torch.float64, device=self.b.device)
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ig_mi_reward(self.b.h, masks.data_ptr(), world.data_ptr(), masks.shape[0],
                                           out.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ig_mi_reward")
        return out

    def next_pose(self, poses, actions, world, radius):
        poses = self._t(poses, torch.float64, (-1, 3))
        actions = self._t(actions, torch.float64, (-1, 2))
        world = self._t(world, torch.int32, (-1,))
        radius = self._t(radius, torch.float64, (-1,))
        Q = poses.shape[0]
        nxt = torch.empty((Q, 3), dtype=torch.float64, device=self.b.device)
        ok = torch.empty((Q,), dtype=torch.uint8, device=self.b.device)
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ig_next_pose(self.b.h, poses.data_ptr(), actions.data_ptr(), world.data_ptr(),
                                           radius.data_ptr(), Q, self.xdt, self.dt, nxt.data_ptr(), ok.data_ptr(),
                                           self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ig_next_pose")
        return nxt, ok

    def rollouts(self, pose0, observed0, exclude, world, n_steps, radius, nsims, seed, max_steps=None,
                 want_observed=False):
        """nsims random roll-outs per query; returns (rewards [Q,nsims], actions [Q,nsims,H], final_pose) and, with
        want_observed, the cells observed along each roll-out [Q,nsims,60]."""
        pose0 = self._t(pose0, torch.float64, (-1, 3))
        Q = pose0.shape[0]
        observed0 = self._t(observed0, torch.int64, (Q, 60))
        exclude = self._t(exclude, torch.int64, (Q, 60))
        world = self._t(world, torch.int32, (Q,))
        n_steps = self._t(n_steps, torch.int32, (Q,))
        radius = self._t(radius, torch.float64, (Q,))
        H = int(max_steps if max_steps is not None else int(n_steps.max().item()) if Q else 0)
        rew = torch.empty((Q, nsims), dtype=torch.float64, device=self.b.device)
        acts = torch.full((Q, nsims, max(H, 1)), 255, dtype=torch.uint8, device=self.b.device)
        fin = torch.empty((Q, nsims, 3), dtype=torch.float64, device=self.b.device)
        obs_out = torch.empty((Q, nsims, 60), dtype=torch.int64, device=self.b.device) if want_observed else None
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ig_rollouts(self.b.h, pose0.data_ptr(), observed0.data_ptr(), exclude.data_ptr(),
                                          world.data_ptr(), n_steps.data_ptr(), radius.data_ptr(), Q, int(nsims),
                                          max(H, 1), self.xdt, self.dt, self.fov, self.range, int(seed),
                                          rew.data_ptr(), acts.data_ptr(), fin.data_ptr(),
                                          None if obs_out is None else obs_out.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ig_rollouts")
        return (rew, acts, fin, obs_out) if want_observed else (rew, acts, fin)


def find_targets_in_obs(other_agents_states, detect_range=5.0):
    """Detector emulation of ig_mcts.find_targets_in_obs (ig_mcts.py:135-152) on an OAS table [.., K, 10]:
    a target is a row of type 1 (Static) within range; the FOV test of the reference compares radians with
    degrees and is therefore always true.  Returns (mask [.., K], global offsets rows[.., 0:2])."""
    oas = other_agents_states
    r = torch.sqrt(oas[..., 0] ** 2 + oas[..., 1] ** 2)
    return (oas[..., 9] == 1.0) & (r <= detect_range), oas[..., 0:2]


class InfoGainBackend(object):
    """numpy-in / numpy-out adapter of InfoGain for dmcts.DecMCTSPlanner (host tree, device primitives)."""

    def __init__(self, ig):
        self.ig = ig

    def next_pose(self, poses, prim_idx, world, radius):
        acts = PRIMITIVES[np.asarray(prim_idx)]
        nxt, ok = self.ig.next_pose(poses, acts, world, radius)
        torch.cuda.synchronize(self.ig.b.device)
        return nxt.cpu().numpy(), ok.cpu().numpy().astype(bool)

    def visible_cells(self, poses, world):
        if len(poses) == 0:
            return np.zeros((0, 60), dtype=np.uint64)
        m = self.ig.visible_cells(poses, world)
        return m.cpu().numpy().view(np.uint64)

    def rollouts(self, pose0, observed0, exclude, world, n_steps, radius, nsims, seed):
        H = int(max(1, np.max(n_steps)))
        rew, acts, fin, obs = self.ig.rollouts(pose0, np.ascontiguousarray(observed0).view(np.int64),
                                               np.ascontiguousarray(exclude).view(np.int64), world, n_steps, radius,
                                               nsims, seed, max_steps=H, want_observed=True)
        return rew.cpu().numpy(), acts.cpu().numpy(), obs.cpu().numpy().view(np.uint64)
