"""GPU (needs a library built with -DTAMF_TIMELINE, see tools/ab_build.sh / TAMF_LIB_OVERRIDE): shader-clock stamps of the
clip-tile GEMM's X wave 0 and Y wave 4, K-tile intervals 4..11 of every workgroup's first tile.
   clip_timeline.py prec epi N K [variant]     (M = 13312)
stamps per interval: X: 0 start, 1 fragments requested + DMA issued, 2 MFMAs issued, 3 after the barrier
                     Y: 0 start, 1 MFMAs done, 2 fragment reads done, 3 after the barrier"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
from kbench import bench, lib
torch.zeros(1, device="cuda")
prec, epi, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
v = int(sys.argv[5], 0) if len(sys.argv) > 5 else 65536
ms = bench(prec, epi, v, 13312, N, K, 3)
buf = np.zeros(512 * 2 * 8 * 4, np.uint64)
rc = lib().tamf_debug_timeline(2, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
b = buf.reshape(512, 2, 8, 4)[:256].astype(np.int64)
print(f"{prec} epi{epi} N={N} K={K}: {ms*1e3:.1f} us")
for h, name, labels in ((0, "X wave 0", ("reads+DMA issue", "MFMAs issued", "barrier wait")), (1, "Y wave 4", ("MFMAs done", "reads done", "barrier wait"))):
    d = np.diff(b[:, h], axis=2)  # [wg][interval][3]
    ok = (b[:, h, :, 0] > 0).all(axis=1)
    d = d[ok]
    tot = b[ok][:, h, 1:, 0] - b[ok][:, h, :-1, 0]
    print(f"{name}: {ok.sum()} workgroups; interval length median {np.median(tot):.0f} cycles (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
    for i, l in enumerate(labels):
        print(f"    {l:18s} median {np.median(d[:, :, i]):7.0f}  p10 {np.percentile(d[:, :, i],10):7.0f}  p90 {np.percentile(d[:, :, i],90):7.0f}")
