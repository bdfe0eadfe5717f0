"""GPU (needs a library built with TAMF_HIPCC_FLAGS=-DTAMF_TIMELINE): per-workgroup timeline of the attention kernel"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import numpy as np, torch
from oakink2_tamf_amd import hip_backend as hb
prec = sys.argv[1]
qkv = torch.randn(64, 201, 3 * 512, device="cuda")
for _ in range(3):
    out = hb.test_attention(prec, qkv, 4)
torch.cuda.synchronize()
n = 512
buf = np.zeros(n * 4, np.uint64)
rc = hb.lib().tamf_debug_timeline(1, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
b = buf.reshape(n, 4)
t0 = b[:, 0].min()
st = (b[:, 0] - t0) / 100.0; pro = (b[:, 1] - b[:, 0]) / 100.0; loop = (b[:, 2] - b[:, 1]) / 100.0; en = (b[:, 2] - t0) / 100.0
print("start us: min %.1f med %.1f max %.1f" % (st.min(), np.median(st), st.max()))
print("prologue us: min %.1f med %.1f max %.1f" % (pro.min(), np.median(pro), pro.max()))
print("loop us: min %.1f med %.1f max %.1f" % (loop.min(), np.median(loop), loop.max()))
print("end us: min %.1f med %.1f max %.1f" % (en.min(), np.median(en), en.max()))
cu = ((b[:, 3] >> 32) << 8) | ((b[:, 3] >> 8) & 0xFF)
u, c = np.unique(cu, return_counts=True)
print("distinct CUs", len(u), "WGs per CU histogram", dict(zip(*np.unique(c, return_counts=True))))
for k in range(0, 16):
    print(k, "cu", hex(int(cu[k])), "start %.1f pro %.1f loop %.1f end %.1f" % (st[k], pro[k], loop[k], en[k]))
