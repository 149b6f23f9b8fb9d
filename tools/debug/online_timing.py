"""evaluate.py's ONLINE branch for one sequence of T frames (evaluate.py:62-64: T + 5 forward_online calls): wall time of the calls
one by one (what rounds 1-4 did) and of mp_stream_replay (round 5), and the latency of a single S = 1 tick (live_demo.py:238-241).
  python tools/debug/online_timing.py [T]     (GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
x = torch.from_numpy(synthetic.make_imu(1, T, seed=61)[0]).cuda()
feed = torch.cat((x, x[-1:].expand(5, -1)))
with MobilePoserNet.from_numpy(sd, smpl) as net:
    net.set_lstm_mode(1)
    for f in feed[:20]:
        net.forward_online(f)
    torch.cuda.synchronize()
    lat = []
    for f in feed[20:220]:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.forward_online(f)
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    print("S = 1 tick (forward_online, 45-frame window): median %.3f ms, p95 %.3f ms, min %.3f ms" % (np.median(lat), np.percentile(lat, 95), lat.min()))
    net.reset_all(); net.last_lfoot_pos, net.last_rfoot_pos = net.feet_pos[0], net.feet_pos[1]   # (reset() keeps them, net.py:84-88)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = [net.forward_online(f) for f in feed]
    torch.cuda.synchronize(); t_ticks = time.perf_counter() - t0
    net.reset_all(); net.last_lfoot_pos, net.last_rfoot_pos = net.feet_pos[0], net.feet_pos[1]
    # (warm-up with the same frames: the workspaces of the chunk shapes exist afterwards -- the first replay of a new shape maps
    #  up to 1.7 GB, 60-90 ms; forward_online_replay cuts a sequence into chunks of 1024, 512, ... frames so that every sequence
    #  finds them)
    net.forward_online_replay(feed); net.reset_all(); net.last_lfoot_pos, net.last_rfoot_pos = net.feet_pos[0], net.feet_pos[1]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = net.forward_online_replay(feed)
    torch.cuda.synchronize(); t_rep = time.perf_counter() - t0
    d = max(float((torch.stack([o[i] for o in a]) - b[i]).abs().max()) for i in (0, 2, 3))
    print("ONLINE branch, T = %d (+5): %d forward_online calls %.1f ms (%.3f ms each); mp_stream_replay %.1f ms (%.1f x); max |difference| %.2e"
          % (T, T + 5, 1e3 * t_ticks, 1e3 * t_ticks / (T + 5), 1e3 * t_rep, t_ticks / t_rep, d))
    net.timing_enable(True)
    net.reset_all(); net.forward_online_replay(feed[:1024]); torch.cuda.synchronize()     # (one chunk: the counters are per library call)
    names = {0: "gemm", 1: "bi256", 4: "bi512", 5: "uni", 6: "foot", 2: "ik", 3: "whole"}
    print("replay of one 1024-frame chunk, classes (launches, ms):", {names[c]: (net.timing_read(c)[0], round(net.timing_read(c)[1], 2)) for c in names})
    others = []
    for n in (T - 100, T - 223, T - 7, T - 1001):                  # other lengths, each for the first time: the chunk shapes are there
        net.reset_all()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.forward_online_replay(feed[:n])
        torch.cuda.synchronize(); others.append((n, 1e3 * (time.perf_counter() - t0)))
    print("replays of other lengths, first time each: " + ", ".join("%d frames %.1f ms" % o for o in others))
    xo = torch.from_numpy(synthetic.make_imu(1, T, seed=61)).cuda()
    net.timing_enable(False)
    for _ in range(3):
        net.reset_all(); net.forward_offline(xo, [T])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        net.forward_offline(xo, [T])
    torch.cuda.synchronize()
    print("forward_offline, one sequence of %d frames: %.3f ms" % (T, 1e2 * (time.perf_counter() - t0)))
    assert net.device_error() == 0
