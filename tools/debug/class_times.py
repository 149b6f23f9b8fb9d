import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
w = synthetic.make_weights(0); smpl = synthetic.synthetic_smpl()
names = {0: "gemm", 1: "bi256", 4: "bi512", 5: "uni", 6: "foot", 2: "ik", 3: "whole"}
m = MobilePoserNet.from_numpy(w, smpl)
m.set_recovery(False)
T = 125
for B in [int(b) for b in sys.argv[1:]] or [256, 512, 1024]:
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
    for k in range(3):
        m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(10):
        m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    m.timing_enable(True)
    m.reset_all(); m.forward_offline(x, [T] * B); torch.cuda.synchronize()
    out = {names[c]: (m.timing_read(c)[0], round(m.timing_read(c)[1], 3)) for c in names}
    m.timing_enable(False)
    print("B=%d  %.3f ms/step  %.2f M frames/s   classes (launches, ms): %s" % (B, dt * 1e3, B * T / dt / 1e6, out), flush=True)
m.close()
