import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import os
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'x6_emulation.py')).read().split("smpl=synthetic")[0])
def f16(x): return np.asarray(x,np.float32).astype(np.float16).astype(np.float32)
def split16(x, n):
    ps=[]; r=np.asarray(x,np.float32)
    for _ in range(n):
        p=f16(r); ps.append(p); r=(r-p).astype(np.float32)
    return ps
WS=16.0
def mm_h(a, wT, terms, ws):
    A=split16(a,2); W=split16(wT*np.float32(ws),2)
    acc=np.zeros((a.shape[0], wT.shape[1]), np.float64)
    for (i,j) in terms: acc += A[i].astype(np.float64)@W[j].astype(np.float64)
    return (acc/ws).astype(np.float32)
def make(terms, ws):
    def f(a, wT, npc, t): return mm_h(a, wT, terms, ws)
    return f
import types
smpl=synthetic.synthetic_smpl()
B,T=16,125
for prof in ("init","trained"):
    sd=synthetic.make_weights(0,prof); imu=synthetic.make_imu(B,T,seed=1); L=[T]*B
    def run(dt, mode, mm=None):
        global MODE, mm_split
        old=mm_split
        if mm is not None: mm_split=mm
        MODE=mode; O.F32=dt
        try:
            n=O.OracleNet(sd,smpl["J"]); p,j,v,c=n.forward(imu,L)
            return dict(r6d=np.asarray(n._last_r6d,np.float64),j=np.asarray(j,np.float64),v=np.asarray(v,np.float64),c=np.asarray(c,np.float64))
        finally: O.F32=np.float32; MODE=None; mm_split=old
    t=run(np.float64,None); f=run(np.float32,None)
    print(prof,"fp32",{k:"%.2e"%np.abs(f[k]-t[k]).max() for k in t})
    for name,terms,ws in (("f16x2 3 products, no scale",[(0,0),(0,1),(1,0)],1.0),("f16x2 4 products, no scale",[(0,0),(0,1),(1,0),(1,1)],1.0),("f16x2 4 products, W x16",[(0,0),(0,1),(1,0),(1,1)],16.0),("f16x2 3 products, W x16",[(0,0),(0,1),(1,0)],16.0)):
        r=run(np.float32,(2,terms),make(terms,ws))
        print(prof,name,{k:"%.2e"%np.abs(r[k]-t[k]).max() for k in t})

# ---- h exchanged with lo's LSB stolen for the tag (lo rounded to 9 mantissa bits), only for the recurrent operand
def lo9(lo):
    u=np.ascontiguousarray(lo.astype(np.float16)).view(np.uint16).astype(np.uint32)
    u=(u+((u>>1)&1))&0xfffe
    return u.astype(np.uint16).view(np.float16).astype(np.float32)
def mm_h_tag(a, wT, terms, ws, is_h):
    A=split16(a,2)
    if is_h: A[1]=lo9(A[1])
    W=split16(wT*np.float32(ws),2)
    acc=np.zeros((a.shape[0], wT.shape[1]), np.float64)
    for (i,j) in terms: acc += A[i].astype(np.float64)@W[j].astype(np.float64)
    return (acc/ws).astype(np.float32)
def make_tag(terms, ws):
    def f(a, wT, npc, t): return mm_h_tag(a, wT, terms, ws, a.shape[0]==B)   # h operand: [B, H]; x: [B*T, K]
    return f
for prof in ("trained",):
    sd=synthetic.make_weights(0,prof); imu=synthetic.make_imu(B,T,seed=1); L=[T]*B
    t=run(np.float64,None)
    terms=[(0,0),(0,1),(1,0)]
    r=run(np.float32,(2,terms),make_tag(terms,16.0))
    print(prof,"f16x2 3 products, W x16, h lo 9 bits",{k:"%.2e"%np.abs(r[k]-t[k]).max() for k in t})
