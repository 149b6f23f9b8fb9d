"""Each RNN block alone (linear1 + two LSTM layers + linear2), B x T, fp32 mode: what the blocks cost without each other."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 125
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
for mod, width in (("joints", 60), ("pose", 132), ("velocity", 132), ("foot_contact", 132)):
    x = torch.randn(B, T, width, device="cuda") * 0.3
    for _ in range(5): net.rnn_forward(mod, x, [T] * B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): net.rnn_forward(mod, x, [T] * B)
    torch.cuda.synchronize()
    print("%-13s alone: %.3f ms" % (mod, (time.perf_counter() - t0) * 20))
