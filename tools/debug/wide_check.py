"""Side-by-side schedule of pose / velocity / foot-contact for small batches: same bits, less time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
shapes = ((1, 3000), (1, 125), (16, 125), (64, 125), (128, 125), (256, 125))
res = {}
for wide in (0, 1):
    os.environ["MP_VARIANT"] = "wide=%d" % wide
    net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
    for mode in (1, 3):
        net.set_lstm_mode(mode)
        for (B, T) in shapes:
            x = torch.from_numpy(synthetic.make_imu(B, T, seed=B + T)).cuda()
            L = [T] * B
            net.reset_all()
            o = [t.clone() for t in net.forward_offline(x, L)]
            o += [t.clone() for t in net.forward_offline(x, L)]
            n = 20 if T < 1000 else 5
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): net.reset_all(); net.forward_offline(x, L)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            assert net.device_error() == 0
            res[(wide, mode, B, T)] = (o, dt)
    net.close()
for mode in (1, 3):
    for (B, T) in shapes:
        a, ta = res[(0, mode, B, T)]; b, tb = res[(1, mode, B, T)]
        same = all(torch.equal(u, v) for u, v in zip(a, b))
        worst = max(float((u - v).abs().max()) for u, v in zip(a, b))
        # (64 < B <= 128, fp32: the side-by-side schedule runs the pose layers on 8 instead of 16 slices -- other kernels, equal to 5e-6)
        print("mode %d  %4d x %4d: serial %7.3f ms  side-by-side %7.3f ms  bitwise equal: %s (max abs difference %.1e)"
              % (mode, B, T, ta * 1e3, tb * 1e3, same, worst))
