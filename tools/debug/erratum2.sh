#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/err
export MP_LIB_PATH=$PWD/mobileposer_amd/libmp_pk_all.so
for g in 0 1; do for mode in 3 1; do
  echo "graph=$g" | tee -a gpurun_out/err/probe2.log
  MP_GRAPH=$g GPU_MAX_HW_QUEUES=8 timeout 600 python tools/debug/erratum.py $mode 300 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/err/probe2.log
done; done
# the round-1 detector as it was: whole forward_offline (IK + FK + solver beside the layers), bitwise run-to-run
MP_GRAPH=1 GPU_MAX_HW_QUEUES=8 timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/err/probe2.log
import sys, torch
sys.path.insert(0, '.')
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
net = MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl())
x = torch.from_numpy(synthetic.make_imu(256, 125, seed=41)).cuda()
for mode in (3, 1):
    net.set_lstm_mode(mode)
    ref, bad = None, 0
    for it in range(400):
        net.reset_all()
        o = [t.clone() for t in net.forward_offline(x, [125] * 256)]
        if ref is None: ref = o
        else: bad += sum(int((a.view(torch.int32) != b.view(torch.int32)).sum()) for a, b in zip(ref, o))
    print("soak forward_offline graph mode=%d: %d differing words in 400 runs" % (mode, bad))
PY
