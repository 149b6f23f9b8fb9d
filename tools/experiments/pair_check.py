"""mp_lstm_pair (two slabs per workgroup) against mp_lstm_fused: bitwise equality of every output, and timing."""
import os, sys, time, subprocess, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet

def run(mask, stagger=None):
    os.environ["MP_PAIR"] = str(mask)
    if stagger: os.environ["MP_PAIR_MODE"] = stagger
    net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
    outs = {}
    rng = np.random.default_rng(5)
    for (B, T) in ((256, 125), (300, 20), (40, 50), (17, 3), (1024, 30)):
        x = torch.from_numpy(synthetic.make_imu(B, T, seed=B + T)).cuda()
        L = [int(v) for v in rng.integers(1, T + 1, size=B)]; L[0] = T
        net.reset_all()
        o = [t.clone() for t in net.forward_offline(x, L)]
        o += [t.clone() for t in net.forward_offline(x, L)]     # carried velocity state
        assert net.device_error() == 0, (B, T)
        outs[(B, T)] = o
    x = torch.from_numpy(synthetic.make_imu(256, 125, seed=1)).cuda()
    L = [125] * 256
    for _ in range(5): net.reset_all(); net.forward_offline(x, L)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): net.reset_all(); net.forward_offline(x, L)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    net.close()
    return outs, dt

ref, t_ref = run(0)
print("MP_PAIR=0: %.3f ms per 256x125 forward_offline" % (t_ref * 1e3))
for mask, st in ((1, "0"), (1, "1"), (2, "0"), (3, "1")):
    got, t = run(mask, st)
    bad = 0
    for k in ref:
        for a, b in zip(ref[k], got[k]):
            if not torch.equal(a, b): bad += 1; print("  differs", k, float((a - b).abs().max()))
    print("MP_PAIR=%d: %.3f ms, %d differing tensors" % (mask, t * 1e3, bad))
