"""HBM rate of the kinematics kernels at the BASELINE size (N = 32 000 frames), HIP events over 50 launches each.
   python tools/debug/kin_bw.py        (GPU box; MP_VARIANT=kin_scalar=1 for the scalar-access kernels)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
N = 32000
r6d = torch.randn(N, 96, device="cuda")
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
import ctypes as C
from mobileposer_amd.net import _ptr
pose = torch.empty(N, 24, 3, 3, device="cuda"); Rg = torch.empty_like(pose); jg = torch.empty(N, 24, 3, device="cuda")
st = net._stream()
t_ik = timed(lambda: net._lib.mp_reduced_global_to_full(net._h, _ptr(r6d), N, _ptr(pose), st))
t_fk = timed(lambda: net._lib.mp_fk(net._h, _ptr(pose), None, N, _ptr(Rg), _ptr(jg), st))
print("variant %s" % os.environ.get("MP_VARIANT", "(default)"))
print("mp_r6d_ik  N=%d: %.1f us  -> %.2f TB/s (%.1f MB)" % (N, t_ik, N * (384 + 864) / t_ik * 1e-6, N * (384 + 864) / 1e6))
print("mp_fk      N=%d: %.1f us  -> %.2f TB/s (%.1f MB)" % (N, t_fk, N * (864 + 1152) / t_fk * 1e-6, N * (864 + 1152) / 1e6))
