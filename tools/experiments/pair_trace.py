"""Timeline of the two slabs of workgroup 0 of mp_lstm_pair over steps 64..71 (MP_PERSIST_PROF build path)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MP_PERSIST_PROF"] = "1"
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
B, T = 256, 125
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
for _ in range(3):
    net.rnn_forward("joints", x, [T] * B)
torch.cuda.synchronize()
n = 4096 + 2 * 8 * 8
buf = (C.c_longlong * n)()
net._lib.mp_debug_read_prof(net._h, buf, n)
a = np.array(buf[4096:]).reshape(2, 8, 8)
t0 = a[0, 0, 0]
names = ["top", "xproj_end", "valid", "hmfma_end", "gate", "published"]
for sl in range(2):
    for k in range(8):
        print("slab %d step %d: " % (sl, 64 + k) + "  ".join("%s %6d" % (names[i], a[sl, k, i] - t0) for i in range(6)))
