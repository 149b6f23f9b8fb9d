"""What float64 accumulation of the gate pre-activations in the one-sequence kernels (mp_set_accumulation(h, 64), round 6) costs and
buys on the reference's own call shape: forward_offline of ONE sequence, trained-regime weights (run on the GPU box).
  python tools/debug/acc64.py        -> times for 1 x 125 / 1 x 3000 / 2 x 3000 / 4 x 3000 / one-stream tick / 3000-frame replay,
                                        and the distance from the float64 result at 1 x 3000 on the input of tools/accuracy.py
                                        (seed 1) beside the fp32 oracle's."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from mobileposer_amd import synthetic                      # noqa: E402
from mobileposer_amd.net import MobilePoserNet             # noqa: E402
from oracle import ensemble as ENS                         # noqa: E402

smpl = synthetic.synthetic_smpl()


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with MobilePoserNet.from_numpy(synthetic.make_weights(0), smpl) as net:
    rows = []
    for B, T in ((1, 125), (1, 3000), (2, 3000), (4, 3000)):
        x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
        t = {}
        for bits in (32, 64):
            net.set_accumulation(bits)
            t[bits] = timed(lambda: (net.reset_all(), net.forward_offline(x, [T] * B)), 20 if T < 1000 else 5)
        rows.append("forward_offline %d x %4d: %8.3f ms fp32, %8.3f ms float64 accumulation (%+.1f %%)" % (B, T, 1e3 * t[32], 1e3 * t[64], 100 * (t[64] / t[32] - 1)))
    frames = torch.from_numpy(synthetic.make_imu(1, 3000, seed=2)[0]).cuda()
    t = {}
    for bits in (32, 64):
        net.set_accumulation(bits)
        t[bits] = timed(lambda: (net.reset_all(), net.reset(), net.forward_online_replay(frames)), 3)
    rows.append("replay of 3000 frames:     %8.3f ms fp32, %8.3f ms float64 accumulation (%+.1f %%)" % (1e3 * t[32], 1e3 * t[64], 100 * (t[64] / t[32] - 1)))
    t = {}
    for bits in (32, 64):
        net.set_accumulation(bits)
        net.reset_all(); net.reset()
        k = [0]

        def tick():
            net.forward_online(frames[k[0] % 3000]); k[0] += 1
        t[bits] = timed(tick, 200)
    rows.append("one-stream tick:           %8.3f ms fp32, %8.3f ms float64 accumulation (%+.1f %%)" % (1e3 * t[32], 1e3 * t[64], 100 * (t[64] / t[32] - 1)))
    print("\n".join(rows))

# accuracy at 1 x 3000, trained-regime weights, input seed 1 (the draw of profiles/r06_accuracy_1x3000.json)
T = 3000
sd = synthetic.make_weights(0, profile="trained")
imu = synthetic.make_imu(1, T, seed=1)
truth = ENS.offline_outputs(sd, smpl["J"], imu, T, dtype=np.float64)
ora = ENS.distance(ENS.offline_outputs(sd, smpl["J"], imu, T, dtype=np.float32), truth)
print("1 x 3000, trained-regime weights, input seed 1: max / mean |x - f64|")
print("  numpy fp32 oracle          " + "  ".join("%s %.1e/%.1e" % (k, v[0], v[1]) for k, v in ora.items()))
with MobilePoserNet.from_numpy(sd, smpl) as net:
    import ctypes as C
    for bits in (32, 64):
        net.set_accumulation(bits)
        net.reset_all()
        x = torch.from_numpy(imu).cuda()
        pose, joints, vel, contact, r6d = net.forward(x, [T], return_r6d=True)
        tran = torch.empty(1, T, 3, device="cuda")
        net.translate_offline_into(joints, vel.reshape(1, T, 72), contact, (C.c_int32 * 1)(T), tran)
        got = {"r6d": r6d.cpu().numpy().reshape(T, 96), "joints": joints.cpu().numpy().reshape(T, 72), "vel": vel.cpu().numpy().reshape(T, 72),
               "contact": contact.cpu().numpy().reshape(T, 2), "tran": tran.cpu().numpy().reshape(T, 3)}
        d = ENS.distance(got, truth)
        print("  one-sequence kernels, %2d    " % bits + "  ".join("%s %.1e/%.1e" % (k, v[0], v[1]) for k, v in d.items()))
