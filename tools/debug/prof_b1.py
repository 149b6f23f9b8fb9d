"""Phase counters of ONE layer launch at B = 1 (mp_lstm_v1, or mp_lstm_u8 with MP_VARIANT=vec=0) inside forward_offline:
  python tools/debug/prof_b1.py <module id: 0 joints 1 pose 3 velocity> <layer 0|1> [T]      (s_memtime counts)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MP_PERSIST_PROF"] = "1"
os.environ["MP_PERSIST_PROF_MODULE"] = sys.argv[1]
os.environ["MP_PERSIST_PROF_LAYER"] = sys.argv[2]
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
x = torch.from_numpy(synthetic.make_imu(1, T, seed=1)).cuda()
for _ in range(3):
    net.reset_all(); net.forward_offline(x, [T])
torch.cuda.synchronize()
buf = (C.c_longlong * (512 * 8))()
net._lib.mp_debug_read_prof(net._h, buf, 512 * 8)
a = np.array(buf[:]).reshape(512, 8)
a = a[a[:, 5] > 0]
names = ["x part", "wait for h", "h part", "reduce", "cell+publish"]
print("MP_VARIANT=%r module %s layer %s: workgroups %d, steps %d" % (os.environ.get("MP_VARIANT", ""), sys.argv[1], sys.argv[2], len(a), a[0, 5]))
for xcc in sorted(set(a[:, 7] & 15)):
    sel = a[(a[:, 7] & 15) == xcc]
    per = sel[:, :5] / sel[:, 5:6]
    print("XCC %d (%d workgroups): total %.1f cycles/step; " % (xcc, len(sel), per.sum(axis=1).mean())
          + ", ".join("%s %.1f" % (n, per[:, i].mean()) for i, n in enumerate(names))
          + "; polls/step %.2f; link waits %.1f" % ((sel[:, 6] / sel[:, 5]).mean(), (sel[:, 7] >> 16).mean()))
