"""Launch timeline of one fused LSTM layer (PROF instantiation, s_memrealtime): entry skew of the workgroups, when the weights
are requested, when step 0 starts, when the time loop ends.  usage: launch_timeline.py <module> <layer>"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MP_PERSIST_PROF"] = "1"
os.environ["MP_PERSIST_PROF_LAYER"] = sys.argv[2]
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
B = 256
mod, layer = sys.argv[1], int(sys.argv[2])
if True:
  for T in (25, 125):
    xin = torch.randn(B, T, 132, device="cuda") * 0.3
    for _ in range(3):
        net.rnn_forward(mod, xin, [T] * B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf = (C.c_longlong * (4096 + 512 * 4))()
    net._lib.mp_debug_read_prof(net._h, buf, 4096 + 512 * 4)
    ph = np.array(buf[:4096]).reshape(512, 8)
    sel = ph[:, 5] == T
    a = np.array(buf[4096:]).reshape(512, 4)[sel]
    xcc = (ph[sel, 7] & 0xFF).astype(int)
    t0 = a[:, 0].min()
    us = lambda v: v / 100.0
    print("%s L%d T=%d: %d WGs; entry skew max %.1f us; weight loads issued at mean %.1f (max %.1f); loop start mean %.1f (max %.1f); loop end mean %.1f (min %.1f max %.1f); per-step %.3f us"
          % (mod, layer, T, len(a), us(a[:, 0].max() - t0), us((a[:, 1] - t0).mean()), us((a[:, 1] - t0).max()),
             us((a[:, 2] - t0).mean()), us((a[:, 2] - t0).max()), us((a[:, 3] - t0).mean()), us((a[:, 3] - t0).min()), us((a[:, 3] - t0).max()),
             us((a[:, 3] - a[:, 2]).mean()) / T), flush=True)
    dur = (a[:, 3] - a[:, 2]) / 100.0
    print("   loop duration by XCC:", {int(x): round(float(dur[xcc == x].mean()), 1) for x in sorted(set(xcc))})
