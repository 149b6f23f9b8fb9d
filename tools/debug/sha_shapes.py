"""sha1 of every output of forward_offline (two calls each: zero, then carried velocity state) and of 3 streaming ticks for a
list of shapes -- run under two builds of the library (MP_LIB_PATH) and diff the printouts: bit-identical or not."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
prof = sys.argv[1] if len(sys.argv) > 1 else "init"
net = MobilePoserNet.from_numpy(synthetic.make_weights(0, profile=prof), synthetic.synthetic_smpl())
def sha(ts):
    h = hashlib.sha1()
    for t in ts: h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]
rng = np.random.default_rng(5)
for B, T in ((256, 125), (256, 1), (256, 2), (256, 3), (257, 100), (200, 30), (128, 125), (100, 30), (72, 40), (48, 50), (17, 33), (1, 300), (1024, 20), (512, 45)):
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=B + T)).cuda()
    L = [T] * B
    for b in range(0, B, 7): L[b] = int(rng.integers(1, T + 1))
    L[B - 1] = T
    net.reset_all()
    a = sha(net.forward_offline(x, L)); b = sha(net.forward_offline(x, L))
    print(B, T, a, b, net.device_error())
S = 300
net.reset_all()
net.stream_create(S)
fr = synthetic.make_imu(S, 4, seed=9)
out = []
for k in range(4):
    out.append(sha(net.stream_step(torch.from_numpy(np.ascontiguousarray(fr[:, k])).cuda())) )
print("stream", out)
