"""Phase breakdown (MP_PERSIST_PROF) of ONE module's layer inside the full forward: the other blocks run beside it as they do
in production.   python tools/debug/prof_forward.py <module id: 0 joints 1 pose 2 foot 3 velocity> <layer 0|1> [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MP_PERSIST_PROF"] = "1"
os.environ["MP_PERSIST_PROF_MODULE"] = sys.argv[1]
os.environ["MP_PERSIST_PROF_LAYER"] = sys.argv[2]
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
B, T = (int(sys.argv[3]) if len(sys.argv) > 3 else 256), 125
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
for _ in range(3):
    net.reset_all(); net.forward_offline(x, [T] * B)
torch.cuda.synchronize()
buf = (C.c_longlong * (512 * 8))()
net._lib.mp_debug_read_prof(net._h, buf, 512 * 8)
a = np.array(buf[:]).reshape(512, 8)
a = a[a[:, 5] > 0]
names = ["x-proj mfma", "validate/wait", "h mfma", "reduce", "cell+publish"]
print("module", sys.argv[1], "layer", sys.argv[2], "workgroups:", len(a), "steps:", a[0, 5])
for i, n in enumerate(names):
    per = a[:, i] / a[:, 5]
    print("%-13s mean %8.1f  min %8.1f  max %8.1f  (100 MHz ticks / step)" % (n, per.mean(), per.min(), per.max()))
print("total/step    mean %8.1f ticks = %.2f us" % ((a[:, :5].sum(axis=1) / a[:, 5]).mean(), (a[:, :5].sum(axis=1) / a[:, 5]).mean() / 100))
print("slow-path steps per workgroup: mean %.1f of %d" % (a[:, 6].mean(), a[0, 5]))
# (word 7: bit 8 = all slices of my cluster on my XCD, bit 9 = the partner cluster too (wavefront), bits 16.. = steps that waited for layer 0)
print("all-local clusters: %d of %d, link-local: %d; link waits per workgroup: mean %.1f max %d" % (
    int(((a[:, 7] >> 8) & 1).sum()), len(a), int(((a[:, 7] >> 9) & 1).sum()), (a[:, 7] >> 16).mean(), int((a[:, 7] >> 16).max())))
if len(sys.argv) > 4:        # the two pseudo-directions of a wavefront launch apart (cluster = slab * 2 + layer: XCD x runs clusters 4x .. 4x+3)
    blk = np.nonzero(np.array(buf[:]).reshape(512, 8)[:, 5] > 0)[0]
    layer = ((blk >> 3) // 8) & 1
    for L in (0, 1):
        sel = a[layer == L]
        print("layer %d: total/step %.1f ticks; x-proj %.1f validate %.1f h %.1f reduce %.1f cell %.1f; link waits %.1f" % (
            (L, (sel[:, :5].sum(axis=1) / sel[:, 5]).mean()) + tuple((sel[:, i] / sel[:, 5]).mean() for i in range(5)) + ((sel[:, 7] >> 16).mean(),)))
