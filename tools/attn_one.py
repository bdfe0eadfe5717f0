"""GPU: run the attention kernel alone at the bench shape (for rocprofv3): attn_one.py prec [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import torch
from oakink2_tamf_amd import hip_backend as hb
prec = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
qkv = torch.randn(64, 201, 3 * 512, device="cuda")
for _ in range(iters):
    out = hb.test_attention(prec, qkv, 4)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
