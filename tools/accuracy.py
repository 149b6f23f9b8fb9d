"""Accuracy table for DESIGN.md: how far each implementation of the network forward is from the SAME arithmetic
carried out in float64 (the oracle with its dtype switched), on one seeded batch.

  python tools/accuracy.py [B] [T]      (GPU box; writes gpurun_out/r04_accuracy.json -> profiles/)

Rows: numpy fp32 oracle, torch CPU fp32 (nn.LSTM, what the reference runs), HIP path with exact-fp32 MFMA operands,
HIP path with split-fp16 operands (mode 3; bf16 halves until round 3) -- for BOTH weight profiles of mobileposer_amd.synthetic.make_weights: "init" (uniform
+-1/sqrt(H), gates near 0.5) and "trained" (LSTM weights x 3, forget bias + 1, linear1 x 2: saturated gates, recurrent
gain > 1, long memory -- rounding differences are amplified through the recurrence instead of being forgotten).
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mobileposer_amd import synthetic                      # noqa: E402
from mobileposer_amd.net import MobilePoserNet             # noqa: E402
from oracle import mp_oracle as O                          # noqa: E402
from oracle.torch_ref import TorchNet                      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 125
smpl = synthetic.synthetic_smpl()
imu = synthetic.make_imu(B, T, seed=1)
lengths = [T] * B
MODES = [("HIP, exact fp32 MFMA operands (mode 1)", 1), ("HIP, split-fp16 MFMA operands (mode 3)", 3)]
if os.environ.get("MP_ACCURACY_MODES"):                       # e.g. "1,3,4": further operand modes under test
    MODES = [("HIP, LSTM mode %s" % m, int(m)) for m in os.environ["MP_ACCURACY_MODES"].split(",")]


def run_profile(profile):
    sd = synthetic.make_weights(0, profile=profile)

    def oracle_run(dtype):
        O.F32 = dtype
        try:
            net = O.OracleNet(sd, smpl["J"])
            pose, joints, vel, contact = net.forward(imu, lengths)
            return {"r6d": np.asarray(net._last_r6d, np.float64), "joints": np.asarray(joints, np.float64),
                    "vel": np.asarray(vel, np.float64), "contact": np.asarray(contact, np.float64)}
        finally:
            O.F32 = np.float32

    truth = oracle_run(np.float64)
    rows = {"numpy fp32 oracle": oracle_run(np.float32)}
    tn = TorchNet(sd, smpl["J"])
    tp, tj, tv, tc, tr6 = tn.forward(imu, lengths)
    rows["torch CPU fp32 (nn.LSTM)"] = {"r6d": tr6, "joints": tj, "vel": tv, "contact": tc}
    net = MobilePoserNet.from_numpy(sd, smpl, device="cuda:0")
    x = torch.from_numpy(imu).cuda()
    for name, mode in MODES:
        net.set_lstm_mode(mode)
        net.reset_all()
        pose, joints, vel, contact, r6d = net.forward(x, lengths, return_r6d=True)
        rows[name] = {"r6d": r6d.cpu().numpy(), "joints": joints.cpu().numpy(),
                      "vel": vel.cpu().numpy(), "contact": contact.cpu().numpy()}
    net.close()
    res = {"output_magnitude": {k: float(np.abs(truth[k]).max()) for k in truth}, "max_abs_error": {}, "mean_abs_error": {},
           "p999_abs_error": {}}
    print("weights: %s   (max |output|: %s)" % (profile, {k: "%.2f" % v for k, v in res["output_magnitude"].items()}))
    print("%-42s %10s %10s %10s %10s" % ("max |x - float64|", "r6d", "joints", "velocity", "contact"))
    for name, r in rows.items():
        e = {k: float(np.abs(np.asarray(r[k], np.float64).reshape(truth[k].shape) - truth[k]).max()) for k in truth}
        res["max_abs_error"][name] = e
        print("%-42s %10.2e %10.2e %10.2e %10.2e" % (name, e["r6d"], e["joints"], e["vel"], e["contact"]))
        # (round 5: the max over 3 M outputs of a chaotic recurrence is one unlucky element; mean and 99.9th percentile beside it)
        d = {k: np.abs(np.asarray(r[k], np.float64).reshape(truth[k].shape) - truth[k]) for k in truth}
        res["mean_abs_error"][name] = {k: float(v.mean()) for k, v in d.items()}
        res["p999_abs_error"][name] = {k: float(np.quantile(v, 0.999)) for k, v in d.items()}
        print("%-42s %10.2e %10.2e %10.2e %10.2e" % ("   mean", *[res["mean_abs_error"][name][k] for k in ("r6d", "joints", "vel", "contact")]))
        print("%-42s %10.2e %10.2e %10.2e %10.2e" % ("   99.9 %", *[res["p999_abs_error"][name][k] for k in ("r6d", "joints", "vel", "contact")]))
    return res


out = {"batch": B, "frames": T, "reference": "oracle arithmetic in float64",
       "profiles": {p: run_profile(p) for p in ("init", "trained")}}
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", os.environ.get("MP_ACCURACY_OUT", "r05_accuracy.json")), "w"), indent=1)
