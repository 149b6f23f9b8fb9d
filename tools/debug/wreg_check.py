"""mp_lstm_fused<256,8,KIN,1> (four 512-register waves, AccVGPR weights) against the eight-wave kernel: bits and time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet

def run(mask):
    os.environ["MP_VARIANT"] = "wreg=%d" % mask
    net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
    outs = {}
    rng = np.random.default_rng(5)
    for (B, T) in ((256, 125), (300, 20), (200, 7)):
        x = torch.from_numpy(synthetic.make_imu(B, T, seed=B + T)).cuda()
        L = [int(v) for v in rng.integers(1, T + 1, size=B)]; L[0] = T
        net.reset_all()
        o = [t.clone() for t in net.forward_offline(x, L)]
        o += [t.clone() for t in net.forward_offline(x, L)]
        assert net.device_error() == 0, (B, T)
        outs[(B, T)] = o
    x = torch.from_numpy(synthetic.make_imu(256, 125, seed=1)).cuda()
    L = [125] * 256
    for _ in range(5): net.reset_all(); net.forward_offline(x, L)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): net.reset_all(); net.forward_offline(x, L)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    net.close()
    return outs, dt

ref, t_ref = run(0)
print("wreg=0: %.3f ms per 256x125 forward_offline" % (t_ref * 1e3))
for mask in (1, 2, 3, 0):
    got, t = run(mask)
    bad, mx = 0, 0.0
    for k in ref:
        for a, b in zip(ref[k], got[k]):
            if not torch.equal(a, b): bad += 1; mx = max(mx, float((a - b).abs().max()))
    print("wreg=%d: %.3f ms, %d differing tensors (max abs diff %.2e)" % (mask, t * 1e3, bad, mx))
