"""Timings of the BASELINE.json configs other than the bench workload (run on the GPU box)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet

net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def report(name, frames, sec):
    print("%-62s %9.3f ms  %12.0f frames/s" % (name, sec * 1e3, frames / sec))


B, T = 256, 125
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
lens = [T] * B
report("configs[1] joints module only, 256 x 125", B * T, timeit(lambda: net.rnn_forward("joints", x, lens)))
report("configs[2] forward + FK, 256 x 125 (facade, incl. clones)", B * T,
       timeit(lambda: (net.reset_all(), net.forward_kinematics(net.forward(x, lens)[0]))))
B4 = 1024
x4 = torch.from_numpy(synthetic.make_imu(B4, T, seed=2)).cuda()
l4 = [T] * B4
report("configs[3] forward_offline (net + solver), 1024 x 125, 1 GPU", B4 * T,
       timeit(lambda: (net.reset_all(), net.forward_offline(x4, l4)), reps=5, warm=2))
x1 = torch.from_numpy(synthetic.make_imu(1, 3000, seed=3)).cuda()
report("configs[0]-like single sequence T=3000 forward_offline", 3000,
       timeit(lambda: (net.reset_all(), net.forward_offline(x1, [3000])), reps=3, warm=1))
print("device error:", net.device_error())

# throughput against the batch size (forward_offline = net + FK + solver, T = 125)
for Bs in (16, 64, 128, 256, 512, 1024, 2048):
    xs = torch.from_numpy(synthetic.make_imu(Bs, T, seed=5)).cuda()
    ls = [T] * Bs
    report("forward_offline, %4d x 125" % Bs, Bs * T, timeit(lambda: (net.reset_all(), net.forward_offline(xs, ls)), reps=5, warm=2))
