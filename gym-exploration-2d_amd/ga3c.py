"""GA3C-CADRL policy on device: state-vector kernel (cagym_ga3c_state) + fused forward kernel
(cagym_ga3c_forward: normalisation, LSTM-64, 3 x FC-256, logits, argmax, action table in ONE launch; since round 4 on the
16-bit matrix cores with split fp32 operands, csrc/cagym_ga3c16.h - CAGYM_GA3C=mfma32 / valu select the exact-fp32 kernels).

Replaces policies/GA3CCADRLPolicy.py:34-43 and GA3C_CADRL/network.py:65-98 (TensorFlow 1.15 session.run per
agent) by one batched forward over every GA3C agent of every world.  Weights come from the converted checkpoint
(tools/convert_ga3c_checkpoint.py -> weights/ga3c_cadrl_*.npz), packed in the blob order of include/cagym.h.
`forward_torch` is a plain-torch restatement kept only as a numerics reference for the tests.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import scenarios as sc

HERE = os.path.dirname(os.path.abspath(__file__))
# network.py:125-148
AVG = np.hstack([[0.0], [0.0, 0.0, 1.0, 0.5], np.tile([0.0, 0.0, 0.0, 0.0, 0.5, 0.0, 1.0], 10)]).astype(np.float32)
STD = np.hstack([[1.0], [5.0, 3.14, 1.0, 1.0], np.tile([5.0, 5.0, 1.0, 1.0, 1.0, 5.0, 1.0], 10)]).astype(np.float32)


def action_table():
    """network.Actions (network.py:8-17): 11 rows (speed factor, delta heading)."""
    rows = [(1.0, -np.pi / 6 + k * (np.pi / 12)) for k in range(5)]
    rows += [(0.5, -np.pi / 6 + k * (np.pi / 6)) for k in range(3)]
    rows += [(0.0, -np.pi / 6 + k * (np.pi / 6)) for k in range(3)]
    return np.array(rows, dtype=np.float64)


class GA3CCADRLPolicy(object):
    def __init__(self, benv, checkpoint="iros18", max_observed=None):
        self.b = benv
        self.L = benv.L
        self.L.cagym_ga3c_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.L.cagym_ga3c_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.cagym_ga3c_act_workspace_bytes.argtypes = [C.c_void_p]
        self.L.cagym_ga3c_act_workspace_bytes.restype = C.c_size_t
        self.L.cagym_ga3c_act.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.cagym_ga3c_act.restype = C.c_int
        self.L.cagym_ga3c_load_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.cagym_ga3c_load_weights.restype = C.c_int
        self._work = None
        path = checkpoint if os.path.exists(checkpoint) else os.path.join(HERE, "weights", "ga3c_cadrl_%s.npz" % checkpoint)
        W = np.load(path)
        dev = benv.device
        self.W = {k: torch.from_numpy(np.ascontiguousarray(W[k])).to(dev) for k in W.files}
        self.avg = torch.from_numpy(AVG).to(dev)
        self.std = torch.from_numpy(STD).to(dev)
        self.table = torch.from_numpy(action_table()).to(dev)
        self.max_observed = int(max_observed if max_observed is not None else min(benv.M - 1, 10))
        self.state = torch.zeros((benv.N, benv.M, 76), dtype=torch.float32, device=dev)
        order = ["lstm_kernel", "lstm_bias", "l1_kernel", "l1_bias", "l2_kernel", "l2_bias", "fc1_kernel", "fc1_bias",
                 "logits_kernel", "logits_bias"]
        self.blob = torch.cat([self.W[k].reshape(-1) for k in order]).contiguous()
        assert self.blob.numel() == 170507
        self._idx = None
        self._idx_episode = None
        # the handle caches the packed blob by ADDRESS: a policy built after another one was freed may get the same address back
        self.load_weights()

    def load_weights(self):
        """Tell the handle that `self.blob` was rewritten in place (it caches the blob as matrix-core operand fragments by address)."""
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ga3c_load_weights(self.b.h, self.blob.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ga3c_load_weights")

    def states(self):
        with torch.cuda.device(self.b.device):
            rc = self.L.cagym_ga3c_state(self.b.h, self.max_observed, self.state.data_ptr(), self.b._stream())
        _lib.check(self.L, self.b.h, rc, "cagym_ga3c_state")
        return self.state

    def agent_index(self):
        """Flat indices (world * M + slot) of the agents whose policy id is POLICY_GA3C.  Recomputed (one small
        device->host sync) only when some world has started a new episode since the last call."""
        b = self.b
        ep = b.state()["episode"]
        if self._idx is None or not torch.equal(ep, self._idx_episode):
            status = b.state()["status"]
            flag = (((status >> 8) & 15) == sc.POLICY_GA3C) & ((status & _lib.FLAG_ACTIVE) != 0)
            self._idx = flag.reshape(-1).nonzero(as_tuple=True)[0].to(torch.int32).contiguous()
            self._idx_episode = ep.clone()
        return self._idx

    def forward(self, state_rows=None, agent_idx=None, ext_actions=None, want_probs=False):
        """Fused forward kernel over `agent_idx` rows of `state_rows` ([*, 76]); returns (action_index, probs)."""
        b = self.b
        st = self.state if state_rows is None else state_rows.contiguous()
        idx = self.agent_index() if agent_idx is None else agent_idx.to(torch.int32).contiguous()
        Bn = int(idx.numel())
        act = torch.empty((Bn,), dtype=torch.int32, device=b.device)
        probs = torch.empty((Bn, 11), dtype=torch.float32, device=b.device) if want_probs else None
        with torch.cuda.device(b.device):
            rc = self.L.cagym_ga3c_forward(b.h, self.blob.data_ptr(), st.data_ptr(), idx.data_ptr(), Bn,
                                           None if ext_actions is None else ext_actions.data_ptr(), act.data_ptr(),
                                           None if probs is None else probs.data_ptr(), b._stream())
        _lib.check(self.L, b.h, rc, "cagym_ga3c_forward")
        return act, probs

    def forward_torch(self, x75):
        """Numerics reference only: softmax_p [B, 11] for NN inputs x75 [B, 75] (= state[..., 1:])."""
        W = self.W
        x = x75.float()
        xn = (x - self.avg) / self.std
        B = x.shape[0]
        n = x[:, 0].to(torch.int32)
        h = torch.zeros((B, 64), dtype=torch.float32, device=x.device)
        c = torch.zeros_like(h)
        seq = xn[:, 5:].reshape(B, 10, 7)
        for t in range(10):
            z = torch.cat([seq[:, t], h], dim=1) @ W["lstm_kernel"] + W["lstm_bias"]
            i, j, f, o = z.split(64, dim=1)
            c2 = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
            h2 = torch.sigmoid(o) * torch.tanh(c2)
            live = (t < n).unsqueeze(1)
            c = torch.where(live, c2, c)
            h = torch.where(live, h2, h)
        y = torch.relu(torch.cat([xn[:, 1:5], h], dim=1) @ W["l1_kernel"] + W["l1_bias"])
        y = torch.relu(y @ W["l2_kernel"] + W["l2_bias"])
        y = torch.relu(y @ W["fc1_kernel"] + W["fc1_bias"])
        p = torch.softmax(y @ W["logits_kernel"] + W["logits_bias"], dim=1)
        return (p + 1e-4) / (1.0 + 1e-4 * 11)

    def act(self, ext_actions=None, fused=True):
        """Fill ext_actions [N,M,2] f32 with (pref_speed * a0, a1) for every agent whose policy id is
        POLICY_GA3C (GA3CCADRLPolicy.find_next_action, :34-43); other rows are left untouched.
        fused (default): one cagym_ga3c_act call - selection, state vectors of the selected agents and the network stay on
        the device, no host synchronisation; fused=False: the three-call path (states of every slot, host-side index list)."""
        b = self.b
        if ext_actions is None:
            ext_actions = torch.zeros((b.N, b.M, 2), dtype=torch.float32, device=b.device)
        assert ext_actions.is_contiguous() and ext_actions.dtype == torch.float32
        if not fused:
            self.states()
            self.forward(ext_actions=ext_actions)
            return ext_actions
        if self._work is None:
            self._work = torch.empty((int(self.L.cagym_ga3c_act_workspace_bytes(b.h)),), dtype=torch.uint8, device=b.device)
        with torch.cuda.device(b.device):
            rc = self.L.cagym_ga3c_act(b.h, self.blob.data_ptr(), self.max_observed, self._work.data_ptr(), ext_actions.data_ptr(), b._stream())
        _lib.check(self.L, b.h, rc, "cagym_ga3c_act")
        return ext_actions
