"""Soak: forward_offline over a rotating set of (B, T, ragged lengths) for N seconds on one handle (eager launches, default
schedule for each size, carried velocity state reset each time); every repeat of a shape must reproduce its first result
bitwise; no recovery, no device error.  usage: soak.py [seconds]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
rng = np.random.default_rng(7)
shapes = [(1, 300), (1, 45), (2, 200), (3, 90), (4, 125), (5, 1), (16, 125), (33, 60), (64, 125), (72, 90), (96, 125), (100, 31), (128, 125), (129, 40), (200, 77), (256, 125), (300, 50), (512, 25), (700, 20)]
cases = []
for B, T in shapes:
    L = [int(v) for v in rng.integers(1, T + 1, size=B)]
    L[int(rng.integers(0, B))] = T
    cases.append((B, T, L, torch.from_numpy(synthetic.make_imu(B, T, seed=B + T)).cuda()))
first, runs, t0 = {}, 0, time.time()
while time.time() - t0 < secs:
    k = int(rng.integers(0, len(cases)))
    B, T, L, x = cases[k]
    net.reset_all(); net.velocity.rnn_state = None
    outs = net.forward_offline(x, L)
    outs2 = net.forward_offline(x, L)                       # carried velocity state
    h = hashlib.md5()
    for o in list(outs) + list(outs2):
        h.update(o.cpu().numpy().tobytes())
    d = h.hexdigest()
    if k in first and first[k] != d:
        print("MISMATCH at run", runs, "shape", (B, T)); sys.exit(1)
    first.setdefault(k, d)
    runs += 1
print("soak: %d double forwards over %d shapes in %.0f s, all repeats bitwise identical; recoveries %d, device error %d"
      % (runs, len(first), time.time() - t0, net.recovery_count, net.device_error()))
sys.exit(0 if net.recovery_count == 0 else 2)
