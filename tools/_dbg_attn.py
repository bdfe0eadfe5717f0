import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd import hip_backend as hb
torch.manual_seed(0)
B,S,H,hd=1,21,1,64
def run(q,k,v,prec="f32"):
    qkv=torch.cat([q,k,v],dim=2)
    return hb.test_attention(prec, qkv.cuda(), H).cpu()
q=torch.randn(B,S,hd); k=torch.randn(B,S,hd)
ref=torch.softmax((q.double()@k.double().transpose(1,2))/8.0,-1)[0]   # [query][key]
W=torch.zeros(S,S)
for kk in range(S):
    v=torch.zeros(B,S,hd); v[0,kk,:]=1.0
    W[:,kk]=run(q,k,v)[0,:,0]
torch.set_printoptions(precision=3, linewidth=220, sci_mode=False)
print("effective weights, query 0:", W[0]); print("reference        query 0:", ref[0].float())
print("ratio q0:", (W[0]/ref[0].float()))
print("ratio q5:", (W[5]/ref[5].float()))
print("row sums:", W.sum(1))
