#!/bin/bash
# round 6, seventh GPU call: B = 2..4 on the one-sequence kernels (new test + the round-5 B <= 16 test + small-shape fuzz), timing, profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -q -x -k "few_sequences or single_sequence_kernel or one_frame_calls or starved" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06_seq_tests.txt; tail -25 gpurun_out/r06_seq_tests.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_small_batches.txt
import os, sys, time, subprocess
CHILD = r'''
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
m = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
for B, T in ((1, 125), (2, 125), (3, 125), (4, 125), (5, 125), (16, 125), (1, 3000), (2, 3000), (4, 3000)):
    x = torch.from_numpy(synthetic.make_imu(B, T, seed=1)).cuda()
    for _ in range(5): m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if T < 1000 else 5
    for _ in range(n): m.reset_all(); m.forward_offline(x, [T] * B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("  %2d x %4d: %.3f ms  %.2f M frames/s" % (B, T, 1e3 * dt, B * T / dt / 1e6), flush=True)
for S in (1, 2, 3, 4, 5, 8):
    m.reset_all(); m.stream_create(S)
    f = torch.from_numpy(synthetic.make_imu(S, 200, seed=2)).cuda()
    for k in range(50): m.stream_step(f[:, k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(100): m.stream_step(f[:, 50 + k])
    torch.cuda.synchronize(); print("  tick S = %d: %.3f ms" % (S, 1e3 * (time.perf_counter() - t0) / 100), flush=True)
'''
for v in ("", "vec=0"):
    env = dict(os.environ); env["MP_VARIANT"] = v
    if not v: env.pop("MP_VARIANT")
    print("MP_VARIANT=%r" % v, flush=True)
    print(subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout)
PY
timeout 600 python tools/debug/fuzz_shapes.py 60 small 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r06_fuzz_small.txt
