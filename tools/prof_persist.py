"""Debug helper: phase breakdown of the persistent LSTM kernel (run on the GPU box with MP_PERSIST_PROF=1)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MP_PERSIST_PROF"] = "1"
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
B, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 256), 125
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
mod = sys.argv[1] if len(sys.argv) > 1 else "joints"
xin = x if mod == "joints" else torch.randn(B, T, 132, device="cuda") * 0.3
for _ in range(3):
    net.rnn_forward(mod, xin, [T] * B)
torch.cuda.synchronize()
buf = (C.c_longlong * (512 * 8))()
net._lib.mp_debug_read_prof(net._h, buf, 512 * 8)
a = np.array(buf[:]).reshape(512, 8)
a = a[a[:, 5] > 0]
names = ["x-proj mfma", "validate/wait", "h mfma", "reduce", "cell+publish"]
print("workgroups:", len(a), "steps:", a[0, 5])
for i, n in enumerate(names):
    per = a[:, i] / a[:, 5]
    print("%-13s mean %8.1f  min %8.1f  max %8.1f  (memtime ticks / step; 100 MHz => x10 ns)" % (n, per.mean(), per.min(), per.max()))
tot = a[:, :5].sum(axis=1) / a[:, 5]
print("total/step    mean %8.1f" % tot.mean())

print("slow-path (stale granules at first look) steps per workgroup: mean %.1f of %d" % (a[:, 6].mean(), a[0, 5]))

print("workgroups whose whole slab is on their own XCD (fast L2 transport):", int((a[:, 7] >= 256).sum()), "of", len(a),
      " XCC ids seen:", sorted(set(int(x) & 0xFF for x in a[:, 7])))
