"""Randomised shapes: the default fp32 configuration (velocity wavefront with the foot-contact riders in pose layer 0 / the
wavefront, epoch-tagged exchange, side-by-side / half-chip schedules with host-placed clusters, 16- and 32-slice small-batch
layers) against the plainest configuration of the same library (every layer a launch of its own on 8 or 16 slices, no riders, no
wavefront, memset per launch, serial schedule, round-robin placement) -- two handles in one process, same inputs, repeated calls
with carried velocity state."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
small = len(sys.argv) > 2 and sys.argv[2] == "small"
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
new = MobilePoserNet.from_numpy(sd, smpl)
os.environ["MP_VARIANT"] = "wf=0,vf=0,epoch_tags=0,wide=0,slices16=0,exclusive=0,half=0,slices32=0,vec=0"
old = MobilePoserNet.from_numpy(sd, smpl)
rng = np.random.default_rng(2024)
worst = 0.0
for case in range(n_cases):
    B = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 48, 64, 65, 80, 96, 97, 100, 112, 128, 129, 255, 256, 257, 300, 511, 700])) if case % 3 else int(rng.integers(1, 400))
    T = int(rng.integers(1, 90))
    if small:                                            # the one-sequence kernels and the few-row linear layers: B * T around 128 rows
        B = int(rng.choice([1, 1, 1, 2, 3, 4]))
        T = int(rng.integers(1, 400)) if case % 4 else int(rng.integers(1, 130 // B + 2))
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=1000 + case)).cuda()
    L = [int(v) for v in rng.integers(1, T + 1, size=B)]
    L[int(rng.integers(0, B))] = T
    outs = []
    for net in (new, old):
        net.reset_all()
        o = [t.clone() for t in net.forward_offline(x, L)]
        o += [t.clone() for t in net.forward_offline(x, L)]
        assert net.device_error() == 0
        outs.append(o)
    d = max(float((a - b).abs().max()) for a, b in zip(*outs))
    assert all(torch.isfinite(a).all() for a in outs[0])
    worst = max(worst, d)
    assert d < 2e-5, (case, B, T, d)
print("fuzz%s: %d random (B, T, lengths) cases, default vs plainest configuration: max abs difference %.2e" % (" (B <= 4)" if small else "", n_cases, worst))
new.close(); old.close()
