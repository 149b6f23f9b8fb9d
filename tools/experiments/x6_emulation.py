import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from mobileposer_amd import synthetic
from oracle import mp_oracle as O
def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)
def split(x, n):
    ps=[]; r=np.asarray(x,np.float32)
    for _ in range(n):
        p=bf16(r); ps.append(p); r=(r-p).astype(np.float32)
    return ps
def mm_split(a, wT, npieces, terms):
    A=split(a,npieces); W=split(wT,npieces)
    acc=np.zeros((a.shape[0], wT.shape[1]), np.float64)
    for (i,j) in terms: acc += A[i].astype(np.float64)@W[j].astype(np.float64)
    return acc.astype(np.float32)
X3=[(0,0),(0,1),(1,0)]
X6=[(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]
MODE=None
orig=O._lstm_direction
def lstm_dir(xs, lengths, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse):
    if MODE is None: return orig(xs, lengths, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse)
    npc, terms = MODE
    F32=np.float32
    B,T,_=xs.shape; H=w_hh.shape[1]
    h=h0.astype(F32).copy(); c=c0.astype(F32).copy()
    out=np.zeros((B,T,H),F32); bias=(b_ih+b_hh).astype(F32)
    xproj=mm_split(xs.reshape(B*T,-1), np.ascontiguousarray(w_ih.T), npc, terms).reshape(B,T,4*H)
    whT=np.ascontiguousarray(w_hh.T); rows=np.arange(B)
    for s in range(T):
        active=lengths>s
        if not active.any(): break
        t_idx=np.where(active,(lengths-1-s) if reverse else s,0)
        g=(xproj[rows,t_idx]+mm_split(h,whT,npc,terms)+bias).astype(F32)
        i=O._sigmoid(g[:,0:H]); f=O._sigmoid(g[:,H:2*H]); gg=np.tanh(g[:,2*H:3*H],dtype=F32); o=O._sigmoid(g[:,3*H:])
        c_new=(f*c+i*gg).astype(F32); h_new=(o*np.tanh(c_new,dtype=F32)).astype(F32)
        a=active[:,None]; c=np.where(a,c_new,c); h=np.where(a,h_new,h)
        out[rows[active],t_idx[active]]=h_new[active]
    return out,h,c
O._lstm_direction=lstm_dir
smpl=synthetic.synthetic_smpl()
B,T=16,125
for prof in ("init","trained"):
    sd=synthetic.make_weights(0,prof); imu=synthetic.make_imu(B,T,seed=1); L=[T]*B
    def run(dt, mode):
        global MODE
        MODE=mode; O.F32=dt
        try:
            n=O.OracleNet(sd,smpl["J"]); p,j,v,c=n.forward(imu,L)
            return dict(r6d=np.asarray(n._last_r6d,np.float64),j=np.asarray(j,np.float64),v=np.asarray(v,np.float64),c=np.asarray(c,np.float64))
        finally: O.F32=np.float32; MODE=None
    t=run(np.float64,None); f=run(np.float32,None); x3=run(np.float32,(2,X3)); x6=run(np.float32,(3,X6))
    for name,r in (("fp32",f),("x3",x3),("x6",x6)):
        print(prof,name,{k:"%.2e"%np.abs(r[k]-t[k]).max() for k in t})
