"""evaluate.py feeds sequences of many different lengths: every new (B, T) is a new plan (workspaces).  Wall time per
forward_offline call of ONE sequence when every call has another length, against repeated calls of one length."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
x = torch.from_numpy(synthetic.make_imu(1, 3200, seed=3)).cuda()
with MobilePoserNet.from_numpy(sd, smpl) as net:
    net.set_lstm_mode(1)
    for _ in range(3):
        net.reset_all(); net.forward_offline(x[:, :3000], [3000])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        net.reset_all(); net.forward_offline(x[:, :3000], [3000])
    torch.cuda.synchronize(); same = (time.perf_counter() - t0) / 20
    lens = [2000 + 37 * i for i in range(30)]
    ts = []
    for T in lens:
        xx = x[:, :T].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.reset_all(); net.forward_offline(xx, [T])
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("one length repeated (T = 3000): %.2f ms per call; 30 calls of 30 different lengths (2000..3073): median %.2f ms, mean %.2f, max %.2f"
          % (1e3 * same, np.median(ts), ts.mean(), ts.max()))
    ts2 = []
    for T in lens:                      # the same lengths again: plans still cached?
        xx = x[:, :T].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.reset_all(); net.forward_offline(xx, [T])
        torch.cuda.synchronize(); ts2.append(time.perf_counter() - t0)
    ts2 = np.array(ts2) * 1e3
    print("the same 30 lengths again: median %.2f ms, mean %.2f, max %.2f" % (np.median(ts2), ts2.mean(), ts2.max()))
    assert net.device_error() == 0
