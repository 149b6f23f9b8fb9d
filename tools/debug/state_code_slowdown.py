"""tests/test_gpu_round5.py::test_state_code_does_not_switch_the_placement_tables_off fails now and then on its timing: what runs
in the slow case?  Repeats the scenario and prints the per-class timing of a forward before / after the state-code recovery."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
B, T = 64, 60
imu = torch.from_numpy(synthetic.make_imu(B, T, seed=3)).cuda()
names = {0: "gemm", 1: "bi256", 4: "bi512", 5: "uni", 6: "foot", 7: "per-step", 2: "ik", 3: "whole"}

def ms(net, reps=20):
    for _ in range(3):
        net.reset_all(); net.forward(imu, [T] * B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        net.reset_all(); net.forward(imu, [T] * B)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

def classes(net):
    net.timing_enable(True)
    net.reset_all(); net.forward(imu, [T] * B); torch.cuda.synchronize()
    out = {names[c]: (net.timing_read(c)[0], round(net.timing_read(c)[1], 3)) for c in names}
    net.timing_enable(False)
    return out

for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    with MobilePoserNet.from_numpy(sd, smpl) as net:
        net.set_lstm_mode(1)
        before = ms(net)
        rng = np.random.Generator(np.random.PCG64(9))
        x = torch.from_numpy((rng.standard_normal((B, 4, 132)) * 0.5).astype(np.float32)).cuda()
        h0 = (rng.standard_normal((2, B, 256)) * 0.3).astype(np.float32)
        h0[1, 3, 9] = 2.5
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            net.rnn_forward("velocity", x, [4] * B, (torch.from_numpy(h0).cuda(), torch.from_numpy(np.zeros_like(h0)).cuda()))
        info = net.device_info()
        after = ms(net)
        print("trial %d: before %.3f ms, after %.3f ms; recoveries %d; info %s" % (trial, before, after, net.recovery_count, {k: info[k] for k in info if k != "build_id"}))
        if after > 1.3 * before:
            print("   slow: classes", classes(net))
            print("   again: %.3f ms" % ms(net))
