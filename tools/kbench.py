"""GPU: time individual GEMM configurations through the C-ABI bench hook (A/B of kernel variants in one process)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import torch
from oakink2_tamf_amd.hip_backend import lib, PRECISIONS

def bench(prec, epi, krot, M, N, K, iters=20):
    ms = ctypes.c_float()
    rc = lib().tamf_bench_gemm(PRECISIONS[prec], epi, krot, M, N, K, iters, ctypes.byref(ms), None)
    if rc != 0:
        return float("nan")
    return ms.value

if __name__ == "__main__":
    torch.cuda.init(); torch.zeros(1, device="cuda")
    M = 13312
    shapes = [("qkv", 1, M, 1536, 512), ("ffn1", 0, M, 2048, 512), ("ffn2_plain", 3, M, 512, 2048),
              ("qkv_ln", 11, M, 1536, 512), ("ffn1_ln", 10, M, 2048, 512), ("outproj_res", 12, M, 512, 512), ("ffn2_res", 12, M, 512, 2048)]  # the forms of the step (deferred LayerNorm)
    precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16x3", "bf16", "f32"]
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
    for prec in precs:
        for name, epi, m, n, k in shapes:
            row = []
            for rep in range(2):
                for v in variants:
                    ms = bench(prec, epi, v, m, n, k)
                    row.append(f"k{v}: {ms*1e3:7.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF")
            print(f"{prec:7s} {name:11s} " + " | ".join(row), flush=True)
