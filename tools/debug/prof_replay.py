"""Phase counters of the velocity chain of mp_stream_replay (B = 1, T = 45 N steps, both layers in one wavefront launch):
  python tools/debug/prof_replay.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MP_PERSIST_PROF"] = "1"
os.environ["MP_PERSIST_PROF_MODULE"] = "3"
os.environ["MP_PERSIST_PROF_LAYER"] = "0"
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
frames = torch.from_numpy(synthetic.make_imu(1, N, seed=1)[0]).cuda()
for _ in range(2):
    net.reset_all(); net.forward_online_replay(frames)
torch.cuda.synchronize()
buf = (C.c_longlong * (512 * 8))()
net._lib.mp_debug_read_prof(net._h, buf, 512 * 8)
a = np.array(buf[:]).reshape(512, 8)
a = a[a[:, 5] > 0]
names = ["x part", "wait for h", "h part", "reduce", "cell+publish"]
print("MP_VARIANT=%r replay of %d frames: workgroups %d, steps %d" % (os.environ.get("MP_VARIANT", ""), N, len(a), a[0, 5]))
for xcc in sorted(set(a[:, 7] & 15)):
    sel = a[(a[:, 7] & 15) == xcc]
    per = sel[:, :5] / sel[:, 5:6]
    print("XCC %d (%d workgroups): total %.1f cycles/step; " % (xcc, len(sel), per.sum(axis=1).mean())
          + ", ".join("%s %.1f" % (n, per[:, i].mean()) for i, n in enumerate(names))
          + "; polls/step %.2f; link waits %.1f" % ((sel[:, 6] / sel[:, 5]).mean(), (sel[:, 7] >> 16).mean()))
