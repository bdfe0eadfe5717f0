import os, sys
R='/root/repo' if os.path.isdir('/root/repo/tools') else os.environ.get('GRAFT_REPO_ROOT','.')
sys.path[:0]=[R, os.path.join(R,'tools')]
import torch
from kbench import bench
torch.zeros(1,device='cuda')
M=13312
for prec in ("bf16x3","bf16"):
    for k in (32 if prec!="bf16" else 64, 512):
        ms=min(bench(prec,0,-1,M,2048,k) for _ in range(3))
        print(f"act={os.environ.get('TAMF_BENCH_ACT','2')} {prec:7s} ffn1 K={k:4d} {ms*1e3:7.1f} us", flush=True)
