"""forward_offline of small batches and ticks of a few streams, default configuration against MP_VARIANT=vec=0 (the 32-slice MFMA
kernels for everything below 33 sequences): what the one-sequence kernels and their group variant buy (round 6).
  python tools/debug/small_batches.py"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, torch, numpy as np
sys.path.insert(0, %r)
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
W, S = synthetic.make_weights(0), synthetic.synthetic_smpl()
m = MobilePoserNet.from_numpy(W, S)
for B, T in ((1, 125), (2, 125), (3, 125), (4, 125), (5, 125), (8, 125), (12, 125), (16, 125), (17, 125), (32, 125), (1, 3000), (2, 3000), (4, 3000), (8, 3000), (16, 3000)):
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
    for _ in range(5): m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if T < 1000 else 5
    for _ in range(n): m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("  %%2d x %%4d: %%7.3f ms  %%.2f M frames/s" %% (B, T, 1e3 * dt, B * T / dt / 1e6), flush=True)
m.close()
for Sn in (1, 2, 4, 8, 16, 32):
    with MobilePoserNet.from_numpy(W, S) as m:
        m.stream_create(Sn)
        f = torch.from_numpy(synthetic.make_imu(Sn, 200, seed=2)).cuda()
        for k in range(50): m.stream_step(f[:, k])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(100): m.stream_step(f[:, 50 + k])
        torch.cuda.synchronize(); print("  tick of %%2d streams: %%.3f ms" %% (Sn, 1e3 * (time.perf_counter() - t0) / 100), flush=True)
''' % REPO
for v in ("", "vec=0"):
    env = dict(os.environ)
    env.pop("MP_VARIANT", None)
    if v:
        env["MP_VARIANT"] = v
    print("MP_VARIANT=%r" % v, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(r.stdout + ("".join(l for l in r.stderr.splitlines(True) if "amdgpu.ids" not in l)[-600:] if r.returncode else ""))
