"""Which sequences of a starved call come out NaN / right / wrong (recovery off)?  python tools/debug/starve_rows.py [skip]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MP_WAIT_MS"] = "15"
import numpy as np, torch
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
skip = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, T = 256, 24
x = torch.from_numpy(synthetic.make_imu(B, T, seed=78)).cuda()
with MobilePoserNet.from_numpy(synthetic.make_weights(0), synthetic.synthetic_smpl()) as m:
    m.set_lstm_mode(1); m.set_recovery(False)
    want = [t.clone() for t in m.forward_offline(x, [T] * B)]
    for blk in (0, 9, 77):
        m.reset_all()
        m._lib.mp_debug_drop_workgroup(m._h, blk, skip, 1)
        got = [t.clone() for t in m.forward_offline(x, [T] * B)]
        try: m.finish()
        except RuntimeError as e: print("finish:", str(e)[:60])
        for name, k in (("joints", 1), ("tran", 2), ("contact", 3)):
            g, w = got[k].flatten(1), want[k].flatten(1)
            nan = torch.isnan(g).any(dim=1)
            wrong = (~nan) & ((g - w).abs().max(dim=1).values > 1e-4)
            print("block", blk, name, "NaN rows:", nan.nonzero().flatten().tolist()[:20], "WRONG finite rows:", wrong.nonzero().flatten().tolist()[:20])
