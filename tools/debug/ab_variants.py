"""A/B of MP_VARIANT settings on ONE box: every variant in a process of its own (the library reads MP_VARIANT at create), the
variants interleaved over `rounds` rounds.  Per run: ms per step of mp_forward_offline (+FK) at B x 125 over `steps` steps, the
event-timed GEMM class per forward, and the sha1 of every output (variants that claim bit-identical results must agree).

  python tools/debug/ab_variants.py [B] [rounds] variant [variant ...]        (variant '-' = default)
"""
import hashlib
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import sys, time, os, hashlib, ctypes as C
sys.path.insert(0, %r)
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
B, T, steps = int(sys.argv[1]), 125, int(sys.argv[2])
m = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
f32 = torch.float32
o = [torch.empty(B * T, 24, 3, 3, device="cuda", dtype=f32), torch.empty(B, T, 72, device="cuda", dtype=f32), torch.empty(B, T, 72, device="cuda", dtype=f32),
     torch.empty(B, T, 2, device="cuda", dtype=f32), torch.empty(B, T, 3, device="cuda", dtype=f32), torch.empty(B * T, 24, 3, 3, device="cuda", dtype=f32),
     torch.empty(B * T, 24, 3, device="cuda", dtype=f32)]
lens = (C.c_int32 * B)(*([T] * B))
def step():
    m._lib.mp_reset_state(m._h, 1)
    m.forward_offline_into(x, lens, *o)
for _ in range(20): step()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
sha = hashlib.sha1()
for t in o: sha.update(t.cpu().numpy().tobytes())
m.timing_enable(True); step(); torch.cuda.synchronize()
gemm = m.timing_read(0)
m.timing_enable(False)
print("%%.4f ms/step   gemm class %%d launches %%.4f ms   sha1 %%s" %% (best * 1e3, gemm[0], gemm[1], sha.hexdigest()[:16]))
m.close()
''' % REPO

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    args = sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1].isdigit() else sys.argv[1:]
    rounds = int(args[0]) if args and args[0].isdigit() else 3
    variants = (args[1:] if args and args[0].isdigit() else args) or ["-"]
    for r in range(rounds):
        for v in variants:
            env = dict(os.environ)
            if v != "-":
                env["MP_VARIANT"] = v
            else:
                env.pop("MP_VARIANT", None)
            out = subprocess.run([sys.executable, "-c", CHILD, str(B), "200"], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if "ms/step" in l]
            print("round %d  %-28s %s" % (r, v, line[0] if line else "FAILED: " + out.stderr[-300:]), flush=True)
