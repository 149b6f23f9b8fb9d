"""Randomised shapes, round 4: the split-fp16 mode (mode 3) against the exact-fp32 mode (mode 1) of the same handle, and the
default one-stream full-batch schedule against the round-3 three-stream schedule (MP_VARIANT one_stream=0 -- a second
handle), same inputs, two calls each (carried velocity state), ragged lengths, both weight profiles."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
smpl = synthetic.synthetic_smpl()
rng = np.random.default_rng(4044)
for profile in ("init", "trained"):
    sd = synthetic.make_weights(0, profile=profile)
    os.environ["MP_VARIANT"] = ""
    new = MobilePoserNet.from_numpy(sd, smpl)
    os.environ["MP_VARIANT"] = "one_stream=0,kin_scalar=1"
    old = MobilePoserNet.from_numpy(sd, smpl)
    os.environ["MP_VARIANT"] = ""
    worst_sched, worst_mode = 0.0, 0.0
    for case in range(n_cases):
        B = int(rng.choice([1, 16, 17, 33, 64, 65, 128, 129, 130, 200, 255, 256, 257, 300, 511, 512, 700])) if case % 3 else int(rng.integers(1, 400))
        T = int(rng.integers(1, 90))
        x = torch.from_numpy(synthetic.make_imu(B, T, seed=3000 + case)).cuda()
        L = [int(v) for v in rng.integers(1, T + 1, size=B)]
        L[int(rng.integers(0, B))] = T
        outs = {}
        for name, net, mode in (("new1", new, 1), ("old1", old, 1), ("new3", new, 3)):
            net.set_lstm_mode(mode)
            net.reset_all()
            o = [t.clone() for t in net.forward_offline(x, L)]
            o += [t.clone() for t in net.forward_offline(x, L)]
            assert net.device_error() == 0 and all(torch.isfinite(a).all() for a in o), (name, case, B, T)
            outs[name] = o
        # rows past a sequence's length carry padding semantics in both; compare everything
        d_s = max(float((a - b).abs().max()) for a, b in zip(outs["new1"], outs["old1"]))
        d_m = max(float((a - b).abs().max()) for a, b in zip(outs["new1"], outs["new3"]))
        worst_sched, worst_mode = max(worst_sched, d_s), max(worst_mode, d_m)
        assert d_s == 0.0 or d_s < 2e-5, (profile, case, B, T, d_s)
        assert d_m < (2e-5 if profile == "init" else 2e-3), (profile, case, B, T, d_m)
    print("fuzz (%s weights): %d random (B, T, lengths) cases x 2 calls: one-stream/frag vs round-3 schedule max |diff| %.2e; "
          "mode 3 (split-fp16) vs mode 1 max |diff| %.2e" % (profile, n_cases, worst_sched, worst_mode))
    new.close(); old.close()
