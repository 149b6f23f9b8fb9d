import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
mode = int(sys.argv[1]); reps = int(sys.argv[2])
B, T = 256, 125
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
net.set_lstm_mode(mode)
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
L = [T] * B
bad_total, lanes, cols, runs_bad = 0, np.zeros(64, np.int64), np.zeros(9, np.int64), 0
for r in range(reps):
    net.reset_all()
    pose, joints, vel, contact = net.forward(x, L)
    torch.cuda.synchronize()
    r6d = net._io[(B, T)]["r6d"].clone()
    alone = net._reduced_global_to_full(r6d)
    torch.cuda.synchronize()
    d = (pose.view(-1, 9).view(torch.int32) != alone.view(-1, 9).view(torch.int32))
    n = int(d.any(dim=1).sum())
    if n:
        runs_bad += 1
        idx = d.any(dim=1).nonzero().flatten().cpu().numpy()
        np.add.at(lanes, idx % 64, 1)
        cols += d.sum(dim=0).cpu().numpy()
    bad_total += n
print("OLD REV (packed fp32 on) mode=%d reps=%d: %d bad entries in %d runs" % (mode, reps, bad_total, runs_bad))
if bad_total:
    print(" by lane:", {int(i): int(v) for i, v in enumerate(lanes) if v})
    print(" by output element:", cols.tolist())
