import ctypes as C, os, sys, warnings
os.environ["MP_WAIT_MS"]="15"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet, _ptr
w=synthetic.make_weights(0); smpl=synthetic.synthetic_smpl()
B,T=256,12
rng=np.random.Generator(np.random.PCG64(82))
x=torch.from_numpy((rng.standard_normal((B,T,132))*0.5).astype(np.float32)).cuda()
st0=torch.from_numpy((rng.standard_normal((2,2,B,256))*0.3).astype(np.float32)).cuda()
lens=(C.c_int32*B)(*([T]*B))
m=MobilePoserNet.from_numpy(w,smpl)
m.set_lstm_mode(1)
y_ref=torch.empty(B,T,72,device="cuda"); st_ref=st0.clone()
print(m._lib.mp_rnn_forward(m._h,3,_ptr(x),lens,B,T,_ptr(y_ref),_ptr(st_ref),_ptr(st_ref),m._stream()))
y=torch.empty(B,T,72,device="cuda"); st=st0.clone()
m._lib.mp_debug_drop_workgroup(m._h,8,0,1)
print(m._lib.mp_rnn_forward(m._h,3,_ptr(x),lens,B,T,_ptr(y),_ptr(st),_ptr(st),m._stream()), m.recovery_count)
torch.cuda.synchronize()
print("y nan rows", torch.isnan(y).flatten(1).any(1).nonzero().flatten().tolist()[:20], "st nan", torch.isnan(st).sum().item(),
      [torch.isnan(st[a,b]).any(1).nonzero().flatten().tolist()[:20] for a in range(2) for b in range(2)])
print("diff", float((y-y_ref).abs().nan_to_num(0).max()), float((st-st_ref).abs().nan_to_num(0).max()))
# non-aliased
y2=torch.empty(B,T,72,device="cuda"); sto=torch.empty_like(st0)
m._lib.mp_debug_drop_workgroup(m._h,8,0,1)
print(m._lib.mp_rnn_forward(m._h,3,_ptr(x),lens,B,T,_ptr(y2),_ptr(st0),_ptr(sto),m._stream()), m.recovery_count)
print("non-aliased: y nan", torch.isnan(y2).sum().item(), "st nan", torch.isnan(sto).sum().item(), float((y2-y_ref).abs().nan_to_num(0).max()))
