import torch, sys
sys.path.insert(0, "/root/repo")
from p2pfl_amd.ops import hip as h
import torch.nn.functional as F
torch.manual_seed(0)
DEV="cuda"
def route(g, code, P):
    K,B,C,_ = g.shape
    out = torch.zeros(K,B,C,2*P,2*P,device=DEV)
    gg = g.view(K,B,C,P,P); cc = code.view(K,B,C,P,P)
    for q in range(4):
        out[..., (q>>1)::2, (q&1)::2] = torch.where(cc==q, gg, torch.zeros_like(gg))
    return out
for ks in (5,3):
  for cores in (False, True):
    for K,B in ((3,32),(4,16)):
        Cin,Cout=32,64
        w = torch.randn(K,Cout,Cin*ks*ks,device=DEV)*0.05
        g = torch.randn(K,B,Cout,49,device=DEV)
        code = torch.randint(0,5,(K,B,Cout,49),device=DEV).to(torch.uint8)
        da1 = torch.empty(K,B,Cin,196,device=DEV)
        h.conv2_dgrad(g,code,w,da1,ks,coresident=cores,prec="fp32-mfma")
        dpre = route(g,code,7)
        worst=0
        for k in range(K):
            wk = w[k].reshape(Cout,Cin,ks,ks)
            ref = torch.nn.grad.conv2d_input((B,Cin,14,14), wk, dpre[k], padding=ks//2).reshape(B,Cin,196)
            err = (da1[k]-ref).abs()
            rel = (da1[k]-ref).norm()/ref.norm()
            worst=max(worst, rel.item())
            pix = err.amax(dim=(0,1))
            bad = (pix > 1e-3).nonzero().flatten().tolist()
        print(ks, cores, K, B, "rel", worst, "bad pixels", bad[:20])
