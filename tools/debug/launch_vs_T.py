import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
w = synthetic.make_weights(0); smpl = synthetic.synthetic_smpl()
names = {0: "gemm", 1: "bi256", 4: "bi512", 5: "uni", 6: "foot", 2: "ik", 3: "whole"}
m = MobilePoserNet.from_numpy(w, smpl)
m.set_recovery(False)
B = 256
for T in (5, 25, 50, 125, 250):
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
    for k in range(3):
        m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize()
    acc = {}
    for rep in range(5):
        m.timing_enable(True)
        m.reset_all(); m.forward_offline(x, [T] * B); torch.cuda.synchronize()
        for c in names:
            n, ms = m.timing_read(c)[:2]
            acc.setdefault(names[c], []).append(ms / max(n, 1))
        m.timing_enable(False)
    print("T=%d per-launch us:" % T, {k: round(1e3 * min(v), 1) for k, v in acc.items()}, flush=True)
m.close()
