"""GPU: the attention kernel alone (tamf_bench_attention) under ablation bits / kernel selections:
    python tools/attn_bench.py [precs] [abl list] [tuning list] [B]
abl (needs a -DTAMF_BENCH build, TAMF_LIB_OVERRIDE): 1 no LDS-DMA, 2 no MFMAs, 4 no fragment reads, 8 no exp2 / split, 16 no store;
tuning: -1 = resident-K kernel, 0x200fffff = streaming kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401
import torch
from oakink2_tamf_amd.hip_backend import lib, PRECISIONS

precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16x3", "bf16"]
abls = [int(v, 0) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
tunes = [int(v, 0) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
torch.zeros(1, device="cuda")
L = lib()
L.tamf_bench_attention.argtypes = [ctypes.c_int32] * 8 + [ctypes.c_void_p, ctypes.c_void_p]
for prec in precs:
    for tune in tunes:
        row = []
        for rep in range(2):
            for abl in abls:
                ms = ctypes.c_float()
                rc = L.tamf_bench_attention(PRECISIONS[prec], B, 201, 4, 128, 50, abl, tune, ctypes.byref(ms), None)
                row.append(f"abl {abl:2d}: {ms.value * 1e3:6.1f} us" if rc == 0 else f"abl {abl}: rc {rc}")
        print(f"{prec:7s} B={B} tuning {tune:#x}: " + " | ".join(row), flush=True)
