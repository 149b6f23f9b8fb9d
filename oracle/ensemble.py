"""An ensemble of LEGAL fp32 evaluations of the same network: the oracle with the K order of every dot product permuted.

TEST INFRASTRUCTURE ONLY (see oracle/mp_oracle.py).  Why it exists (round 6): on trained-regime weights a sequence of thousands
of frames amplifies rounding so much that the maximum distance of ONE fp32 evaluation from the float64 result is a heavy-tailed
draw -- permuting the summation order of the numpy oracle alone moves its own maximum by up to 10 x at [1, 3000, 60]
(profiles/r06_b1_precision_emulation.txt).  "k x the oracle's distance" on such a shape compares two draws; the ensemble gives
the band fp32 implementations occupy there (tests/golden/make_golden.py records its statistics in golden G17 next to the
reference's own outputs, tests/test_gpu_round6.py holds the HIP path to that band).

The arithmetic is the oracle's (`mp_oracle._lstm_direction`, models/rnn.py:27 restated): only the order in which the terms of
`x @ W_ih^T` and `h @ W_hh^T` are added differs -- any BLAS / any kernel picks one such order.
"""
import numpy as np

from . import mp_oracle as O


def permuted_direction(rng):
    """A drop-in for mp_oracle._lstm_direction whose dot products run over a random permutation of K (fp32 throughout)."""
    F32 = np.float32

    def direction(xs, lengths, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse):
        B, T, _ = xs.shape
        H = w_hh.shape[1]
        h = h0.astype(F32).copy()
        c = c0.astype(F32).copy()
        out = np.zeros((B, T, H), dtype=F32)
        p1, p2 = rng.permutation(w_ih.shape[1]), rng.permutation(H)
        bias = (b_ih + b_hh).astype(F32)
        xproj = (xs.reshape(B * T, -1)[:, p1] @ np.ascontiguousarray(w_ih[:, p1].T)).reshape(B, T, 4 * H).astype(F32)
        w_hh_t = np.ascontiguousarray(w_hh[:, p2].T)
        rows = np.arange(B)
        for s in range(T):
            active = lengths > s
            if not active.any():
                break
            t_idx = np.where(active, (lengths - 1 - s) if reverse else s, 0)
            g = (xproj[rows, t_idx] + h[:, p2] @ w_hh_t + bias).astype(F32)
            i = O._sigmoid(g[:, 0 * H:1 * H])
            f = O._sigmoid(g[:, 1 * H:2 * H])
            gg = np.tanh(g[:, 2 * H:3 * H], dtype=F32)
            o = O._sigmoid(g[:, 3 * H:4 * H])
            c_new = (f * c + i * gg).astype(F32)
            h_new = (o * np.tanh(c_new, dtype=F32)).astype(F32)
            a = active[:, None]
            c = np.where(a, c_new, c)
            h = np.where(a, h_new, h)
            out[rows[active], t_idx[active]] = h_new[active]
        return out, h, c
    return direction


OUTPUTS = ("r6d", "joints", "vel", "contact", "tran")


def offline_outputs(sd, J, imu, T, dtype=np.float32, perm_seed=None):
    """forward_offline of ONE sequence through the oracle -> dict of float64 arrays (r6d [T,96], joints [T,72], vel [T,72],
    contact [T,2], tran [T,3]).  dtype=np.float64: the same arithmetic carried out exactly (the yardstick);
    perm_seed: an ensemble member (fp32, permuted summation order)."""
    keep = O._lstm_direction
    O.F32 = dtype
    if perm_seed is not None:
        O._lstm_direction = permuted_direction(np.random.Generator(np.random.PCG64(perm_seed)))
    try:
        net = O.OracleNet(sd, J)
        pose, joints, tran, contact = net.forward_offline(np.asarray(imu).reshape(1, T, 60), [T])
        vel = net._last_vel
        return {"r6d": np.asarray(net._last_r6d, np.float64).reshape(T, 96), "joints": np.asarray(joints, np.float64).reshape(T, 72),
                "vel": np.asarray(vel, np.float64).reshape(T, 72), "contact": np.asarray(contact, np.float64).reshape(T, 2),
                "tran": np.asarray(tran, np.float64).reshape(T, 3)}
    finally:
        O.F32 = np.float32
        O._lstm_direction = keep


def distance(got, truth):
    """{output: (max, mean)} of |got - truth| over all frames."""
    res = {}
    for k in OUTPUTS:
        d = np.abs(np.asarray(got[k], np.float64).reshape(truth[k].shape) - truth[k])
        res[k] = (float(d.max()), float(d.mean()))
    return res
