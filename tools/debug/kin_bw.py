"""The kinematics kernels at the BASELINE size (N = 32 000 frames), 50 calls each through the C ABI.
   cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p -o kin -- python tools/debug/kin_bw.py    (GPU box) -> KERNEL durations: the HBM rates of
                                                                          DESIGN.md section 4 are bytes / those (profiles/r06_kin_bw.txt)
   python tools/debug/kin_bw.py        prints the time per CALL (HIP events around the loop): ~30 us of enter / launch / leave per call on the host,
                                       three times a kernel's duration -- the call rate, not the kernel's
   (MP_VARIANT=kin_scalar=1 for the scalar-access kernels)"""
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
print("per CALL (host-bound; kernel durations: rocprofv3, see the header):")
print("mp_r6d_ik  N=%d: %.1f us  -> %.2f TB/s (%.1f MB)" % (N, t_ik, N * (384 + 864) / t_ik * 1e-6, N * (384 + 864) / 1e6))
print("mp_fk      N=%d: %.1f us  -> %.2f TB/s (%.1f MB)" % (N, t_fk, N * (864 + 1152) / t_fk * 1e-6, N * (864 + 1152) / 1e6))
loc = torch.empty_like(pose)
t_gl = timed(lambda: net._lib.mp_inverse_kinematics_r(net._h, _ptr(Rg), N, _ptr(loc), st))
print("mp_global_to_local N=%d: %.1f us  -> %.2f TB/s (%.1f MB)" % (N, t_gl, N * (864 + 864) / t_gl * 1e-6, N * (864 + 864) / 1e6))
