import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viforsdes_amd import _hip
dev = torch.device('cuda:0')
def run(B,T,S,C,P,H,L, iters=20):
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g)*sc).to(dev)
    NO = S + S*(S+1)//2
    ws = [rn(3*H,S+C+P,sc=.08), rn(3*H,H,sc=.12), rn(3*H,sc=.1), rn(3*H,sc=.1), rn(L-1,3*H,H,sc=.12), rn(L-1,3*H,H,sc=.12), rn(L-1,3*H,sc=.1), rn(L-1,3*H,sc=.1), rn(NO,H,sc=.1), torch.ones(NO).to(dev)]
    x0, ctx, theta, eps = rn(B,S), rn(B,T+1,C).to(torch.bfloat16)[:, :-1], rn(B,P).abs(), rn(B,T,S)
    gp, gm, gl = rn(B,T+1,S), rn(B,T,S), rn(B,T,S,S)
    dt=0.1
    for mode in ('eval','train','bwd'):
        for it in range(3):
            out = _hip.head_forward(x0,ctx,theta,eps,ws,dt,mode!='eval')
            if mode=='bwd': _hip.head_backward(gp,gm,gl,ctx,theta,eps,out[0],out[3],out[4],ws,dt)
        torch.cuda.synchronize(); t0=time.time()
        for it in range(iters):
            if mode!='bwd': out = _hip.head_forward(x0,ctx,theta,eps,ws,dt,mode!='eval')
            else: _hip.head_backward(gp,gm,gl,ctx,theta,eps,out[0],out[3],out[4],ws,dt)
        torch.cuda.synchronize(); dtm=(time.time()-t0)/iters
        print(f"B={B} T={T} S={S} C={C} H={H} L={L} {mode}: {dtm*1e3:.3f} ms  ({B*T/dtm/1e6:.1f} M path-steps/s)")
run(512,400,2,256,3,64,2)
run(128,100,1,256,3,64,2)
run(256,1000,8,512,16,64,2, iters=5)
