import sys, os, math, torch
sys.path.insert(0, os.getcwd())
from musediffusion_amd import _lib
_lib.use_debug_library()
L = _lib.lib(); DEV="cuda"
def rnd(*s, seed, scale=1.0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))*scale
def to_panel(w):
    r,k=w.shape; return w.bfloat16().reshape(r,k//32,32).permute(1,0,2).contiguous().to(DEV)
def from_panel(p): return p.permute(1,0,2).reshape(p.shape[1],-1).float().cpu()
for M,N,K in [(512,256,96),(520,256,128),(1024,256,128),(512,512,128),(512,256,512),(16384,2048,512)]:
    print('shape',M,N,K)
    X,W,b=rnd(M,K,seed=11),rnd(N,K,seed=12,scale=1/math.sqrt(K)),rnd(N,seed=13,scale=0.5)
    Xp,Wp,bd=to_panel(X),to_panel(W),b.to(DEV)
    outs={}
    for on in (1,0):
        L.mh_gemm_set_buf_dma(on)
        out=torch.zeros(N//32,M,32,device=DEV,dtype=torch.bfloat16)
        _lib.check(L.mh_gemm_bias_act_ex(Xp.data_ptr(),M,1,Wp.data_ptr(),N,1,bd.data_ptr(),None,0,0,out.data_ptr(),M,1,0,M,N,K,2,1,_lib.current_stream()))
        outs[on]=from_panel(out)
    d=(outs[1]-outs[0]).abs()
    bad=(d>1e-3) | torch.isnan(d)
    print("bad frac", float(bad.float().mean()))
    rows=bad.any(1).nonzero().flatten(); cols=bad.any(0).nonzero().flatten()
    print("bad rows", rows[:40].tolist(), len(rows)); print("bad cols", cols[:40].tolist(), len(cols))
    ref=torch.nn.functional.gelu(X.bfloat16().float()@W.bfloat16().float().T+b)
    print("err default", float((outs[0]-ref).abs().max()), "err buf", float((outs[1]-ref).abs().nan_to_num(9).max()))
    
