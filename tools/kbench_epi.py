import sys
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, os.path.join(R,'tools')]
import torch
from kbench import bench
torch.zeros(1,device='cuda')
M=13312
for prec in ("bf16x3","bf16"):
    for name,epi,n,k in (("qkv_epi",1,1536,64),("ffn1_epi",0,2048,64),("ln_epi",2,512,64),("ffn1_k128",0,2048,128),("ffn1_k256",0,2048,256),("ffn1_k512",0,2048,512),("ffn1_k1024",0,2048,1024),("ln_k512",2,512,512),("ln_k1024",2,512,1024),("ln_k2048",2,512,2048)):
        ms=[bench(prec,epi,-1,M,n,k) for _ in range(2)]
        print(f"{prec:7s} {name:10s} {min(ms)*1e3:7.1f} us", flush=True)
