"""GPU (library built with -DTAMF_TIMELINE, TAMF_LIB_OVERRIDE): which workgroup ids share a CU in the persistent 512-workgroup QKV launch
(measured: b and b + 256), how many tiles each CU therefore gets (224 CUs x 5, 32 x 4) and when the 3-tile / 2-tile workgroups end."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
from kbench import bench, lib
torch.zeros(1, device="cuda")
for rep in range(3):
    ms = bench("f16x3", 1, -1, 13312, 1536, 512, 2)
    n = 512
    buf = np.zeros(n * 5, np.uint64)
    rc = lib().tamf_debug_timeline(0, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    b = buf.reshape(n, 5)
    cu = ((b[:, 4] >> 32) << 8) | ((b[:, 4] >> 8) & 0xFF)
    pairs = {}
    for i, c in enumerate(cu):
        pairs.setdefault(int(c), []).append(i)
    diffs = sorted(abs(v[1] - v[0]) for v in pairs.values() if len(v) == 2)
    import collections
    print("rep", rep, "pair index differences histogram:", collections.Counter(diffs).most_common(8))
    xcc = (b[:, 4] >> 32)
    print("   xcc of wg 0..15:", [int(x) for x in xcc[:16]])
    end = (b[:, 3] - b[:, 0].min()).astype(np.float64) / 100.0
    three = end[:224]; two = end[224:]
    print("   end time: 3-tile WGs med %.1f max %.1f | 2-tile WGs med %.1f max %.1f" % (np.median(three), three.max(), np.median(two), two.max()))
