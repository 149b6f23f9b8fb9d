"""What the bench's step costs when the batch starts and ends in HOST memory (run on the GPU box): never part of bench.py's
`value` (inputs are resident there), noted in DESIGN.md section 5.

One step = the call bench.py times (mp_forward_offline with the FK outputs, 256 x 125).  A caller of the drop-in hands over the
IMU windows (`evaluate.py:56`, `.to(device)`) and takes back what `forward_offline` returns -- pose, joints, translation,
contact (`models/net.py:155`); the FK outputs stay on the device (the evaluator consumes them there).
  serial     H2D -> step -> D2H, one after the other, pinned host buffers
  pipelined  two buffer sets; the H2D of batch k + 1 and the D2H of batch k - 1 are enqueued on copy streams in front of the
             call of batch k (the library call itself blocks the host until its batch is done -- recovery on, the default --
             so everything that should overlap it is enqueued first)
  pageable   the serial form from ordinary (unpinned) numpy memory, what `torch.from_numpy(x).to(device)` does
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet

B, T = 256, 125
dev = torch.device("cuda", 0)
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl(), device=dev)
lib, h = net._lib, net._h
f32 = torch.float32
vp = lambda t: C.c_void_p(t.data_ptr())
lens = (C.c_int32 * B)(*([T] * B))

OUT_SHAPES = {"pose": (B * T, 24, 3, 3), "joints": (B, T, 72), "vel": (B, T, 72), "contact": (B, T, 2), "tran": (B, T, 3),
              "rglob": (B * T, 24, 3, 3), "jglob": (B * T, 24, 3)}
BACK = ("pose", "joints", "tran", "contact")           # what forward_offline returns (net.py:155)


def dev_set():
    d = {"imu": torch.empty(B, T, 60, device=dev, dtype=f32)}
    d.update({k: torch.empty(*s, device=dev, dtype=f32) for k, s in OUT_SHAPES.items()})
    return d


def host_set(pinned=True):
    mk = (lambda *s: torch.empty(*s, dtype=f32).pin_memory()) if pinned else (lambda *s: torch.empty(*s, dtype=f32))
    d = {"imu": mk(B, T, 60)}
    d["imu"].copy_(torch.from_numpy(synthetic.make_imu(B, T, seed=1)))
    d.update({k: mk(*OUT_SHAPES[k]) for k in BACK})
    return d


def step(d, stream):
    lib.mp_reset_state(h, 1)
    rc = lib.mp_forward_offline(h, vp(d["imu"]), lens, B, T, vp(d["pose"]), vp(d["joints"]), vp(d["vel"]), vp(d["contact"]),
                                vp(d["tran"]), vp(d["rglob"]), vp(d["jglob"]), C.c_void_p(stream.cuda_stream))
    if rc:
        raise RuntimeError(lib.mp_last_error(h).decode())


def timeit(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps


main = torch.cuda.current_stream(dev)
D, Hp, Hu = dev_set(), host_set(True), host_set(False)
bytes_in = B * T * 60 * 4
bytes_out = sum(int(np.prod(OUT_SHAPES[k])) * 4 for k in BACK)

resident = timeit(lambda: step(D, main), 100)


def serial(Hs):
    D["imu"].copy_(Hs["imu"], non_blocking=True)
    step(D, main)
    for k in BACK:
        Hs[k].copy_(D[k], non_blocking=True)
    torch.cuda.synchronize(dev)


t_serial = timeit(lambda: serial(Hp), 50)
t_pageable = timeit(lambda: serial(Hu), 20)

# copies alone
t_h2d = timeit(lambda: D["imu"].copy_(Hp["imu"], non_blocking=True), 50)
t_d2h = timeit(lambda: [Hp[k].copy_(D[k], non_blocking=True) for k in BACK], 50)

# pipelined: two device sets, two pinned host sets, copy streams for each direction
Ds, Hs = [D, dev_set()], [Hp, host_set(True)]
s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ev_in = [torch.cuda.Event(), torch.cuda.Event()]       # input k is on the device
ev_done = [torch.cuda.Event(), torch.cuda.Event()]     # batch k is computed
ev_back = [torch.cuda.Event(), torch.cuda.Event()]     # outputs of batch k are on the host (its device set is free again)
for e in ev_back + ev_done:
    e.record(main)
with torch.cuda.stream(s_in):
    Ds[0]["imu"].copy_(Hs[0]["imu"], non_blocking=True)
    ev_in[0].record(s_in)
counter = [0]


def pipelined():
    k = counter[0] & 1
    n = k ^ 1
    counter[0] += 1
    with torch.cuda.stream(s_in):                      # the next batch comes in while this one is computed
        s_in.wait_event(ev_back[n])                    # (its device set must have been copied out)
        Ds[n]["imu"].copy_(Hs[n]["imu"], non_blocking=True)
        ev_in[n].record(s_in)
    main.wait_event(ev_in[k])
    main.wait_event(ev_back[k])                        # (the outputs this set held two batches ago have left)
    step(Ds[k], main)                                  # blocks the host until batch k is done (recovery on)
    ev_done[k].record(main)
    with torch.cuda.stream(s_out):                     # ... and goes out while the next one is computed
        s_out.wait_event(ev_done[k])
        for name in BACK:
            Hs[k][name].copy_(Ds[k][name], non_blocking=True)
        ev_back[k].record(s_out)


t_pipe = timeit(pipelined, 100, warm=10)
err = net.device_error()
frames = B * T
print(json.dumps({
    "workload": "bench.py's step (mp_forward_offline + FK outputs, %d x %d) with the batch starting and ending in host memory" % (B, T),
    "bytes_in": bytes_in, "bytes_out": bytes_out, "returned": list(BACK),
    "resident_ms": round(1e3 * resident, 4), "resident_frames_per_s": round(frames / resident, 1),
    "h2d_ms": round(1e3 * t_h2d, 4), "h2d_gbps": round(bytes_in / t_h2d / 1e9, 1),
    "d2h_ms": round(1e3 * t_d2h, 4), "d2h_gbps": round(bytes_out / t_d2h / 1e9, 1),
    "serial_pinned_ms": round(1e3 * t_serial, 4), "serial_pinned_frames_per_s": round(frames / t_serial, 1),
    "pipelined_pinned_ms": round(1e3 * t_pipe, 4), "pipelined_pinned_frames_per_s": round(frames / t_pipe, 1),
    "serial_pageable_ms": round(1e3 * t_pageable, 4), "serial_pageable_frames_per_s": round(frames / t_pageable, 1),
    "device_error": err, "recoveries": net.recovery_count}))
