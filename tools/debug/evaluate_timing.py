"""evaluate_pose (evaluate.py:39-107) per sequence on synthetic sequences of evaluate.py's shape: where the time of one sequence
goes (forward_offline, the evaluator's FK + skinning + metrics, the ONLINE=1 replay and its evaluator pass).
  [ONLINE=1] python tools/debug/evaluate_timing.py [n_sequences] [T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
from mobileposer_amd import evaluate as E
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 6
T0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(5)
data = []
for i in range(n_seq):
    T = T0 - 137 * i
    imu = torch.from_numpy(synthetic.make_imu(1, T, seed=200 + i)[0])
    pose = torch.from_numpy(rng.standard_normal((T, 144)).astype(np.float32))
    joint = torch.zeros(T, 24, 3)
    tran = torch.from_numpy(np.cumsum(rng.standard_normal((T, 3)).astype(np.float32) * 0.01, axis=0))
    data.append((imu, pose, joint, tran))
with MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl()) as net:
    net.set_lstm_mode(1)
    for rep in range(2):                     # the second pass finds the workspaces of every chunk shape
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = E.evaluate_pose(net, data, evaluate_tran=False, verbose=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("pass %d: evaluate_pose over %d sequences of %d..%d frames (ONLINE=%s): %.1f ms = %.1f ms per sequence"
              % (rep, n_seq, data[-1][0].shape[0], T0, os.getenv("ONLINE", "0"), 1e3 * dt, 1e3 * dt / n_seq))
    # the pieces, one sequence
    x = data[0][0].cuda()
    def timed(f, reps=5):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / reps, r
    t_fwd, (pose_p, _j, tran_p, _c) = timed(lambda: (net.reset(), net.forward_offline(x.unsqueeze(0), [x.shape[0]]))[1])
    ev = E.PoseEvaluator(net)
    pose_gt = net.r6d_to_rotation_matrix(data[0][1].cuda()).view(-1, 24, 3, 3)
    t_ev, _ = timed(lambda: ev.eval(pose_p, pose_gt, tran_p=tran_p, tran_t=data[0][3].cuda()))
    print("one %d-frame sequence: forward_offline %.2f ms, evaluator (FK + skinning of both poses + 10 metrics) %.2f ms" % (x.shape[0], t_fwd, t_ev))
    assert net.device_error() == 0
