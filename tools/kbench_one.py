"""GPU: run ONE GEMM configuration (for rocprofv3 --pmc): kbench_one.py prec epi variant M N K [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
from kbench import bench
torch.zeros(1, device="cuda")
prec, epi, variant, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3], 0), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
ms = bench(prec, epi, variant, M, N, K, iters)
print(f"{prec} epi{epi} v{variant} {M}x{N}x{K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TF")
