"""Time forward_offline at one shape (fp32 mode), alternating two values of an environment switch on the same box."""
import os, sys, time, subprocess
var, a, b, B, T = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
code = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
B, T = %d, %d
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda(); L = [T] * B
for _ in range(10): net.reset_all(); net.forward_offline(x, L)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): net.reset_all(); net.forward_offline(x, L)
torch.cuda.synchronize(); print("%%.3f ms" %% ((time.perf_counter() - t0) * 10))
''' % (os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), B, T)
for i in range(3):
    for v in (a, b):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        print(var, "=", v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
