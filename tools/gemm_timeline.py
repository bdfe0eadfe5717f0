"""GPU (needs a library built with TAMF_HIPCC_FLAGS=-DTAMF_TIMELINE): per-workgroup timeline of one GEMM launch: gemm_timeline.py prec epi M N K nwg"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
from kbench import bench, lib
torch.zeros(1, device="cuda")
prec, epi, M, N, K, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
ms = bench(prec, epi, int(sys.argv[7]) if len(sys.argv) > 7 else -1, M, N, K, 3)
buf = np.zeros(n * 5, np.uint64)
rc = lib().tamf_debug_timeline(0, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
b = buf.reshape(n, 5)
t0 = b[:, 0].min()
us = lambda x: x.astype(np.float64) / 100.0
st, pro, kl, ep, en = us(b[:, 0] - t0), us(b[:, 1] - b[:, 0]), us(b[:, 2] - b[:, 1]), us(b[:, 3] - b[:, 2]), us(b[:, 3] - t0)
print(f"{prec} epi{epi} {M}x{N}x{K}: {ms*1e3:.1f} us, {n} workgroups")
for name, v in (("start", st), ("prologue", pro), ("kloop", kl), ("epilogue", ep), ("end", en)):
    print("%-9s min %6.1f  p10 %6.1f  med %6.1f  p90 %6.1f  max %6.1f" % (name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
order = np.argsort(st)
# rounds: cluster by start time
for lo, hi in ((0, 512), (512, 1024), (1024, 1536), (1536, n)):
    idx = order[lo:hi]
    if len(idx) == 0: continue
    print("by start rank %4d-%4d: start med %6.1f  pro med %5.1f  kloop med %5.1f  epi med %5.1f  end med %6.1f" % (lo, hi, np.median(st[idx]), np.median(pro[idx]), np.median(kl[idx]), np.median(ep[idx]), np.median(en[idx])))
cu = ((b[:, 4] >> 32) << 8) | ((b[:, 4] >> 8) & 0xFF)
u, c = np.unique(cu, return_counts=True)
print("distinct CUs", len(u), "WGs per CU histogram", dict(zip(*[x.tolist() for x in np.unique(c, return_counts=True)])))
