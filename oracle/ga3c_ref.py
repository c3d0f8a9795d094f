"""fp64 numpy restatement of the GA3C-CADRL forward pass (TEST INFRASTRUCTURE ONLY).

policies/GA3C_CADRL/network.py:65-98 (NetworkVP_rnn), :8-17 (Actions), :125-148 (normalisation) of the
reference; TensorFlow 1.15 is absent, so network outputs are "parity unpinned": the only pin is the known
answer recorded in SURVEY.md 8(c) (reproduced in tests/test_ga3c.py) plus behaviour (agents reach goals).
TF1 LSTMCell conventions: gates (i, j, f, o), forget_bias 1.0, input = concat[x, h].
"""
import numpy as np

AVG = np.hstack([[0.0], [0.0, 0.0, 1.0, 0.5], np.tile([0.0, 0.0, 0.0, 0.0, 0.5, 0.0, 1.0], 10)])
STD = np.hstack([[1.0], [5.0, 3.14, 1.0, 1.0], np.tile([5.0, 5.0, 1.0, 1.0, 1.0, 5.0, 1.0], 10)])


def action_table():
    a = np.mgrid[1.0:1.1:0.5, -np.pi / 6:np.pi / 6 + 0.01:np.pi / 12].reshape(2, -1).T
    a = np.vstack([a, np.mgrid[0.5:0.6:0.5, -np.pi / 6:np.pi / 6 + 0.01:np.pi / 6].reshape(2, -1).T])
    return np.vstack([a, np.mgrid[0.0:0.1:0.5, -np.pi / 6:np.pi / 6 + 0.01:np.pi / 6].reshape(2, -1).T])


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def forward(W, x75):
    """x75: [B, 75] = state[1:].  Returns softmax_p [B, 11] in fp64."""
    x = np.asarray(x75, dtype=np.float64)
    xn = (x - AVG) / STD
    B = x.shape[0]
    n = x[:, 0].astype(np.int64)
    h = np.zeros((B, 64))
    c = np.zeros((B, 64))
    seq = xn[:, 5:].reshape(B, 10, 7)
    K, b = W["lstm_kernel"].astype(np.float64), W["lstm_bias"].astype(np.float64)
    for t in range(10):
        z = np.concatenate([seq[:, t], h], axis=1) @ K + b
        i, j, f, o = np.split(z, 4, axis=1)
        c2 = _sig(f + 1.0) * c + _sig(i) * np.tanh(j)
        h2 = _sig(o) * np.tanh(c2)
        live = (t < n)[:, None]
        c = np.where(live, c2, c)
        h = np.where(live, h2, h)
    y = np.maximum(0, np.concatenate([xn[:, 1:5], h], axis=1) @ W["l1_kernel"].astype(np.float64) + W["l1_bias"])
    y = np.maximum(0, y @ W["l2_kernel"].astype(np.float64) + W["l2_bias"])
    y = np.maximum(0, y @ W["fc1_kernel"].astype(np.float64) + W["fc1_bias"])
    lg = y @ W["logits_kernel"].astype(np.float64) + W["logits_bias"]
    p = np.exp(lg - lg.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    return (p + 1e-4) / (1.0 + 1e-4 * 11)


def _split16(v):
    """fp32 -> (hi, lo) f16 halves as csrc/cagym_ga3c16.h hands operands to the 16-bit matrix cores: hi = f16(v), lo = f16(v - hi)."""
    v = np.asarray(v, dtype=np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def _mm_split(x, Wm):
    """x @ Wm with both operands split: hi*lo + lo*hi + hi*hi, fp32 accumulation (numpy's order, not the matrix core's - this is
    a model of the ARITHMETIC CLASS the split-f16 kernel computes in, for an error budget on the CPU, not a bit-level twin)."""
    xh, xl = _split16(x)
    wh, wl = _split16(Wm)
    return (xh @ wl + xl @ wh) + xh @ wh


def forward_split_f16(W, x75, split=True):
    """The forward pass in the arithmetic of k_ga3c_forward_h16 (fp32 everywhere, matrix products on split f16 operands, fp32
    accumulation, activations clamped to the f16 range); split=False: the same in plain fp32 (the arithmetic of the exact-fp32
    kernels).  Returns softmax_p [B, 11] (float64 view of fp32 results)."""
    f32 = np.float32
    _mm = _mm_split if split else (lambda a, b: (np.asarray(a, f32) @ np.asarray(b, f32)).astype(f32))
    x = np.asarray(x75, dtype=f32)
    xn = ((x - AVG.astype(f32)) / STD.astype(f32)).astype(f32)
    B = x.shape[0]
    n = x[:, 0].astype(np.int64)
    h = np.zeros((B, 64), f32)
    c = np.zeros((B, 64), f32)
    seq = xn[:, 5:].reshape(B, 10, 7)
    K, b = W["lstm_kernel"].astype(f32), W["lstm_bias"].astype(f32)
    sig = lambda v: (f32(1) / (f32(1) + np.exp(-v, dtype=f32))).astype(f32)
    for t in range(10):
        z = (_mm(np.concatenate([seq[:, t], h], axis=1), K) + b).astype(f32)
        i, j, f, o = np.split(z, 4, axis=1)
        c2 = (sig(f + f32(1)) * c + sig(i) * np.tanh(j, dtype=f32)).astype(f32)
        h2 = (sig(o) * np.tanh(c2, dtype=f32)).astype(f32)
        live = (t < n)[:, None]
        c = np.where(live, c2, c)
        h = np.where(live, h2, h)
    relu = lambda v: np.minimum(np.maximum(v, f32(0)), f32(65504)).astype(f32)
    y = relu(_mm(np.concatenate([xn[:, 1:5], h], axis=1), W["l1_kernel"].astype(f32)) + W["l1_bias"].astype(f32))
    y = relu(_mm(y, W["l2_kernel"].astype(f32)) + W["l2_bias"].astype(f32))
    y = relu(_mm(y, W["fc1_kernel"].astype(f32)) + W["fc1_bias"].astype(f32))
    lg = (_mm(y, W["logits_kernel"].astype(f32)) + W["logits_bias"].astype(f32)).astype(np.float64)
    p = np.exp(lg - lg.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    return (p + 1e-4) / (1.0 + 1e-4 * 11)


def find_next_action(W, state76, pref_speed):
    """GA3CCADRLPolicy.find_next_action (policies/GA3CCADRLPolicy.py:34-43) after the state vector: obs[1:] -> predict_p
    -> argmax -> network.Actions row -> [pref_speed * a0, a1], then the env's float32 action table (env.py:289).
    state76 [B, 76], pref_speed [B].  Returns (actions [B, 2] float32-valued, probabilities [B, 11])."""
    st = np.asarray(state76, dtype=np.float64).reshape(-1, 76)
    p = forward(W, st[:, 1:])
    raw = action_table()[np.argmax(p, axis=1)]
    act = np.stack([np.asarray(pref_speed, dtype=np.float64).reshape(-1) * raw[:, 0], raw[:, 1]], axis=1)
    return act.astype(np.float32).astype(np.float64), p
