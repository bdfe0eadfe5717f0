"""GPU: throughput of the h2o distance kernel at the refiner's real scale (T=196, V=778, 2 objects x 8192 points)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd import geometry
B, T, V, nobj, P = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 196, 778, 2, 8192
g = torch.Generator().manual_seed(0)
hv = (torch.randn(B, T, V, 3, generator=g) * 0.1).cuda(); tr = torch.randn(B, nobj, T, 9, generator=g).cuda(); pts = (torch.randn(B, nobj, P, 3, generator=g) * 0.1).cuda()
geometry.multi_object_h2o_dist(hv, tr, pts); torch.cuda.synchronize()
t = time.perf_counter(); n = 3
for _ in range(n): out = geometry.multi_object_h2o_dist(hv, tr, pts)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
pairs = B * T * V * nobj * P
print(f"h2o_dist B={B}: {dt*1e3:.2f} ms  {pairs/dt/1e12:.2f} Tpair/s  {pairs*8/dt/1e12:.1f} TFLOP/s (8 flop/pair; fp32 vector peak 157.3)  {dt/B*1e3:.2f} ms/clip")
pose = torch.randn(64 * 196, 99, generator=g).cuda()
geometry.pose_repr_to_quat(pose); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): geometry.pose_repr_to_quat(pose)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"pose_decode 64x196 frames: {dt*1e6:.1f} us ({pose.numel()*4/dt/1e9:.0f} GB/s read)")
geometry.contact_min_dist(hv, tr, pts); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n): cm = geometry.contact_min_dist(hv, tr, pts)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f"contact_min_dist B={B}: {dt*1e3:.2f} ms  {pairs/dt/1e12:.2f} Tpair/s  contact ratio {geometry.contact_ratio(cm):.3f}")
import numpy as np
sys.path.insert(0, ROOT)
from oracle.fixtures import icosphere
mv, mf = icosphere(3)                       # 1280 faces (a closed MANO hand has 1554)
mv = mv * np.array([0.05, 0.09, 0.03])
q = (torch.rand(200000, 3, generator=g, dtype=torch.float64) - 0.5).cuda() * 0.2
geometry.mesh_contains(mv, mf, q); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): ins = geometry.mesh_contains(mv, mf, q)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"mesh_contains 200k points x {len(mf)} faces: {dt*1e3:.2f} ms ({q.shape[0]*len(mf)/dt/1e9:.1f} G point-triangle tests/s, f64), inside {int(ins.sum())}")
