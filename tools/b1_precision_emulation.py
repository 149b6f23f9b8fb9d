"""CPU emulation (round 6, verdict r5 item 1): how far does ONE sequence of 2000-3000 frames on trained-regime weights drift from the
float64 result under different arithmetic of the LSTM step?

  fp32          the numpy oracle as it is (sgemm accumulation, fp32 state)
  fp32-perm     the same with the K order of every dot product permuted (another legal fp32 summation order: the "draw")
  dot64         gate pre-activations = fp32(float64 dot products + bias): what a kernel that accumulates in float64 produces;
                activations and state in fp32
  dot64-c64     ... and the cell state carried in float64 inside a call
  fp32-kact     fp32 with the KERNELS' activation formulas -- sigma(x) = rcp(1 + exp2(-x log2 e)), tanh(x) = 1 - 2 rcp(exp2(2 x log2 e) + 1),
                every operation rounded to fp32 (mp_lstm_dev.h sigmoidf_ / tanhf_) -- instead of numpy's correctly rounded exp / tanh
  dot64-kact    dot64 with those formulas: what mp_set_accumulation(h, 64) of commit 686b6c5 computed (round 6, built / measured / removed)

Prints max / mean |x - float64| per output for each (T, input seed).  Nothing here runs on the GPU; it decided the design of
mp_lstm_v1's float64 accumulation.   python tools/b1_precision_emulation.py [T ...]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mobileposer_amd import synthetic                      # noqa: E402
from oracle import mp_oracle as O                          # noqa: E402

F32 = np.float32


def make_direction(kind, rng=None):
    kact = kind.endswith("-kact")
    kind = kind[:-5] if kact else kind

    def sig(x):
        if kact:
            e = np.exp2((F32(-1.4426950408889634) * x).astype(F32)).astype(F32)
            return (F32(1) / (F32(1) + e).astype(F32)).astype(F32)
        return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)

    def tanh_(x):
        if kact:
            e = np.exp2((F32(2.8853900817779268) * x).astype(F32)).astype(F32)
            return (F32(1) - (F32(2) * (F32(1) / (e + F32(1)).astype(F32)).astype(F32)).astype(F32)).astype(F32)
        return np.tanh(x, dtype=F32)

    def direction(xs, lengths, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse):
        B, T, _ = xs.shape
        H = w_hh.shape[1]
        h = h0.astype(F32).copy()
        c = c0.astype(np.float64 if kind == "dot64-c64" else F32).copy()
        out = np.zeros((B, T, H), dtype=F32)
        if kind == "fp32-perm":
            p1, p2 = rng.permutation(w_ih.shape[1]), rng.permutation(H)
            bias = (b_ih + b_hh).astype(F32)
            xproj = (xs.reshape(B * T, -1)[:, p1] @ np.ascontiguousarray(w_ih[:, p1].T)).reshape(B, T, 4 * H).astype(F32)
            whh_t = np.ascontiguousarray(w_hh[:, p2].T)
        elif kind == "fp32":
            bias = (b_ih + b_hh).astype(F32)
            xproj = (xs.reshape(B * T, -1) @ np.ascontiguousarray(w_ih.T)).reshape(B, T, 4 * H).astype(F32)
            whh_t = np.ascontiguousarray(w_hh.T)
        else:
            bias = b_ih.astype(np.float64) + b_hh.astype(np.float64)
            xproj = (xs.reshape(B * T, -1).astype(np.float64) @ w_ih.T.astype(np.float64)).reshape(B, T, 4 * H)
            whh_t = np.ascontiguousarray(w_hh.T).astype(np.float64)
        rows = np.arange(B)
        for s in range(T):
            active = lengths > s
            if not active.any():
                break
            t_idx = np.where(active, (lengths - 1 - s) if reverse else s, 0)
            if kind == "fp32-perm":
                g = (xproj[rows, t_idx] + h[:, p2] @ whh_t + bias).astype(F32)
            elif kind == "fp32":
                g = (xproj[rows, t_idx] + h @ whh_t + bias).astype(F32)
            else:
                g = (xproj[rows, t_idx] + h.astype(np.float64) @ whh_t + bias).astype(F32)
            i, f, o = sig(g[:, :H]), sig(g[:, H:2 * H]), sig(g[:, 3 * H:])
            gg = tanh_(g[:, 2 * H:3 * H])
            if kind == "dot64-c64":
                c_new = f.astype(np.float64) * c + i.astype(np.float64) * gg.astype(np.float64)
                h_new = (o * np.tanh(c_new.astype(F32), dtype=F32)).astype(F32)
            else:
                c_new = (f * c + i * gg).astype(F32)
                h_new = (o * tanh_(c_new)).astype(F32)
            a = active[:, None]
            c = np.where(a, c_new, c)
            h = np.where(a, h_new, h)
            out[rows[active], t_idx[active]] = h_new[active]
        return out, h, c.astype(F32)
    return direction


def run(sd, J, imu, T, kind=None, dtype=np.float32, rng=None):
    keep = O._lstm_direction
    O.F32 = dtype
    if kind:
        O._lstm_direction = make_direction(kind, rng)
    try:
        net = O.OracleNet(sd, J)
        pose, joints, vel, contact = net.forward(imu, [T])
        return {"r6d": np.asarray(net._last_r6d, np.float64), "joints": np.asarray(joints, np.float64),
                "vel": np.asarray(vel, np.float64), "contact": np.asarray(contact, np.float64)}
    finally:
        O.F32 = np.float32
        O._lstm_direction = keep


LEVELS = {}

if __name__ == "__main__":
    Ts = [int(a) for a in sys.argv[1:]] or [2000, 2500, 3000]
    smpl = synthetic.synthetic_smpl()
    sd = synthetic.make_weights(0, profile="trained")
    for T in Ts:
        for seed in (1, 2, 3):
            imu = synthetic.make_imu(1, T, seed=seed)
            truth = run(sd, smpl["J"], imu, T, dtype=np.float64)
            rows = {"fp32": run(sd, smpl["J"], imu, T)}
            for k in range(3):
                rows["fp32-perm%d" % k] = run(sd, smpl["J"], imu, T, "fp32-perm", rng=np.random.default_rng(k))
            rows["dot64"] = run(sd, smpl["J"], imu, T, "dot64")
            rows["dot64-c64"] = run(sd, smpl["J"], imu, T, "dot64-c64")
            rows["fp32-kact"] = run(sd, smpl["J"], imu, T, "fp32-kact")
            rows["dot64-kact"] = run(sd, smpl["J"], imu, T, "dot64-kact")
            print("T = %d, input seed %d      max / mean |x - float64|:  r6d | joints | velocity | contact" % (T, seed))
            means = {}
            for name, r in rows.items():
                d = {k: np.abs(r[k].reshape(truth[k].shape) - truth[k]) for k in truth}
                means[name] = {k: d[k].mean() for k in truth}
                print("  %-12s " % name + " | ".join("%.1e / %.1e" % (d[k].max(), d[k].mean()) for k in ("r6d", "joints", "vel", "contact")), flush=True)
            med = {k: np.median([means[n][k] for n in ("fp32", "fp32-perm0", "fp32-perm1", "fp32-perm2")]) for k in truth}
            for name in ("dot64", "dot64-c64", "fp32-kact", "dot64-kact"):
                LEVELS.setdefault(name, []).append([means[name][k] / med[k] for k in ("r6d", "joints", "vel", "contact")])
    print("noise level = geometric mean over the cases of (mean distance / the median fp32 variant's):  r6d | joints | velocity | contact")
    for name, v in LEVELS.items():
        print("  %-12s " % name + " | ".join("%.2f" % x for x in np.exp(np.mean(np.log(np.array(v)), axis=0))))
