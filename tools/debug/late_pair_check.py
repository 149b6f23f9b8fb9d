"""Schedule 4 (MP_VARIANT late_pair, 64 < B <= 128) against schedules 2 / 3: outputs and time per forward_offline."""
import hashlib, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import hashlib, sys, time, torch
sys.path.insert(0, %r)
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
B, T = int(sys.argv[1]), int(sys.argv[2])
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
L = [T - (i * 7) %% (T // 2) for i in range(B)]; L[0] = T
net.reset_all(); net.velocity.rnn_state = None
outs = net.forward_offline(x, L); outs2 = net.forward_offline(x, L)          # second call: carried velocity state
h = hashlib.md5()
for o in list(outs) + list(outs2): h.update(o.cpu().numpy().tobytes())
L = [T] * B
for _ in range(10): net.reset_all(); net.forward_offline(x, L)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): net.reset_all(); net.forward_offline(x, L)
torch.cuda.synchronize()
print("%%s %%.3f ms err=%%d rec=%%d" %% (h.hexdigest()[:12], (time.perf_counter() - t0) * 10, net.device_error(), net.recovery_count))
''' % REPO
for B in [int(b) for b in sys.argv[1:]] or [72, 96, 112, 128]:
    for v in ("late_pair=0", "late_pair=1", "late_pair=0", "late_pair=1"):
        r = subprocess.run([sys.executable, "-c", code, str(B), "125"], env=dict(os.environ, MP_VARIANT=v), capture_output=True, text=True)
        print(B, v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
