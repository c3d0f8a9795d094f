"""Delegating stand-in for the TensorFlow half of GA3C-CADRL (THIS CONTAINER ONLY, fixture generation).

TensorFlow 1.15 is absent and cannot be installed, so the network's arithmetic (GA3C_CADRL/network.py:65-98 as a TF
graph) stays PARITY UNPINNED.  What can be pinned is the reference's own Python around `predict_p`:
GA3CCADRLPolicy.find_next_action (policies/GA3CCADRLPolicy.py:34-43: state vector -> obs[1:] -> predict_p -> argmax ->
network.Actions row -> pref_speed scaling) inside a reference episode.  `NetworkVP_rnn` below takes the place of the
class of the same name in the reference's network module: same constructor arguments, `simple_load` (asserts the
checkpoint the reference asks for is the one the repo's converted weights come from) and `predict_p`, which evaluates
the oracle's numpy restatement (oracle/ga3c_ref.py) and logs every (input, output) pair.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

WEIGHTS = os.path.join(REPO, "gym-exploration-2d_amd", "weights", "ga3c_cadrl_iros18.npz")
LOG = []   # (x [1, 75] float64, p [1, 11] float64) per predict_p call


class NetworkVP_rnn(object):
    def __init__(self, device, model_name, num_actions):
        assert num_actions == 11
        self.device, self.model_name, self.num_actions = device, model_name, num_actions
        self.W = None

    def simple_load(self, filename=None):
        # GA3CCADRLPolicy.initialize_network (:21-32) default: checkpoints/IROS18/network_01900000
        assert filename is not None and filename.endswith(os.path.join("IROS18", "network_01900000")), filename
        assert os.path.exists(filename + ".index"), "the checkpoint the reference names must exist"
        self.W = dict(np.load(WEIGHTS))

    def predict_p(self, x):
        from oracle import ga3c_ref
        assert self.W is not None, "predict_p before simple_load"
        x = np.asarray(x, dtype=np.float64)
        p = ga3c_ref.forward(self.W, x)
        LOG.append((x.copy(), p.copy()))
        return p


def install(network_module):
    network_module.NetworkVP_rnn = NetworkVP_rnn
