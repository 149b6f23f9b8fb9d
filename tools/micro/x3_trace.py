import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ["MP_PERSIST_PROF"] = "1"
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
net.set_lstm_mode(3)
B, T = 256, 125
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
mod = sys.argv[1] if len(sys.argv) > 1 else "joints"
xin = x if mod == "joints" else torch.randn(B, T, 132, device="cuda") * 0.3
for _ in range(3):
    net.rnn_forward(mod, xin, [T] * B)
torch.cuda.synchronize()
NW = 512 * 8 + 2048 * 32 * 8
buf = (C.c_longlong * NW)()
assert net._lib.mp_debug_read_prof(net._h, buf, NW) == 0
tr = np.array(buf[4096:]).reshape(-1, 8, 32, 8)     # [block][wave][step][stamp]
nb = 256
tr = tr[:nb]
names = ["start", "fetch issued", "A ready", "h mma done", "gates ready", "published", "lds issued", "reg chunks done"]
for blk in (0, 135):
    t = tr[blk]                                         # [wave][step][stamp]
    base = t[:, :, 0].min(axis=0)
    print("block", blk, "stamps relative to the earliest start in the workgroup, mean over 32 steps (cycles)")
    for w in range(8):
        rel = (t[w, :, :8] - base[:, None]).mean(axis=0)
        print("  wave %d: " % w + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names, rel)))
    print("  step period:", np.diff(t[0, :, 0]).mean())
