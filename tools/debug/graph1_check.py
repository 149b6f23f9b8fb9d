import os, sys, subprocess
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import sys, torch
sys.path.insert(0, %r)
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
B, T = int(sys.argv[1]), 20
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
net.set_graph_mode(0); net.reset_all(); net.velocity.rnn_state = None
a = [t.clone() for t in net.forward_offline(x, [T] * B)]
net.set_graph_mode(int(sys.argv[2])); net.reset_all(); net.velocity.rnn_state = None
for k in range(3):
    net.reset_all(); net.velocity.rnn_state = None
    b = net.forward_offline(x, [T] * B)
torch.cuda.synchronize()
print("equal:", all(torch.equal(p, q) for p, q in zip(a, b)), "err", net.device_error())
''' % REPO
for B in (128, 96, 64):
    for var in ("late_pair=1", "late_pair=0"):
        for q in ("8", None):
            env = dict(os.environ, MP_VARIANT=var)
            if q: env["GPU_MAX_HW_QUEUES"] = q
            r = subprocess.run([sys.executable, "-c", code, str(B), "1"], env=env, capture_output=True, text=True)
            print(B, var, "GPU_MAX_HW_QUEUES=%s" % q, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1], flush=True)
