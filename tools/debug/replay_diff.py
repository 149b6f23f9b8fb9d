import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
x = torch.from_numpy(synthetic.make_imu(1, T, seed=61)[0]).cuda()
with MobilePoserNet.from_numpy(sd, smpl) as net:
    net.set_lstm_mode(1)
    a = [net.forward_online(f) for f in x]
    a = [torch.stack([o[i] for o in a]) for i in range(4)]
    ha, ca = net.velocity.rnn_state
    net.reset_all(); net.last_lfoot_pos, net.last_rfoot_pos = net.feet_pos[0], net.feet_pos[1]
    b = net.forward_online_replay(x)
    hb, cb = net.velocity.rnn_state
    for name, p, q in zip(("pose", "joints", "root", "contact"), a, b):
        d = (p - q).abs().flatten(1).max(dim=1).values.cpu().numpy()
        bad = np.nonzero(d > 1e-4)[0]
        print("%-8s max %.3e at frame %d; frames over 1e-4: %d, first %s; d[::300] = %s" % (name, d.max(), int(d.argmax()), len(bad), bad[:5], np.array2string(d[::300], precision=1)))
    dv = (a[2][1:] - a[2][:-1]) - (b[2][1:] - b[2][:-1])
    print("per-frame root velocity difference: max %.3e, mean %.3e" % (float(dv.abs().max()), float(dv.abs().mean())))
    print("final velocity state difference h %.3e c %.3e" % (float((ha - hb).abs().max()), float((ca - cb).abs().max())))
