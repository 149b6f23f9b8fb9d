"""Where do the outputs of the velocity wavefront differ between the same-XCD and the forced any-placement transport?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
B, T = 256, 30
x = torch.from_numpy(synthetic.make_imu(B, T, seed=29)).cuda()
net = MobilePoserNet.from_numpy(sd, smpl)
net.set_lstm_mode(1)
runs = []
for remote in (False, True, True, False):
    net.set_transport(remote)
    net.reset_all()
    runs.append([t.clone() for t in net.forward(x, [T] * B)])
names = ["pose", "joints", "vel", "contact"]
for i, j in ((0, 1), (1, 2), (0, 3)):
    for n, a, b in zip(names, runs[i], runs[j]):
        a, b = a.reshape(B, T, -1), b.reshape(B, T, -1)
        d = (a - b).abs()
        if float(d.max()) == 0:
            print("runs %d vs %d: %-8s identical" % (i, j, n)); continue
        rows = (d.flatten(1).max(dim=1).values > 0).nonzero().flatten().tolist()
        tfirst = [(int((d[r].max(dim=1).values > 0).nonzero()[0])) for r in rows[:8]]
        print("runs %d vs %d: %-8s max %.3e, %d rows differ (first: %s), first differing t of those: %s" % (i, j, n, float(d.max()), len(rows), rows[:8], tfirst))
print("device error", net.device_error())
