"""GPU: run ONE library GEMM / attention (for rocprofv3 --pmc beside tools/kbench_one.py):  lib_one.py <f32|bf16|f16> <M> <N> <K> [iters]
   or  lib_one.py <dtype> sdpa <B> <H> <S> <hd> [iters].  torch F.linear (hipBLASLt / rocBLAS) / F.scaled_dot_product_attention, nothing of
this repo's kernels.  Prints the kernel names torch dispatched (from the profiler's own trace, not from here) and a wall time."""
import sys, time
import torch
import torch.nn.functional as F
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1]]
g = torch.Generator().manual_seed(0)
if sys.argv[2] == "sdpa":
    B, H, S, hd = (int(v) for v in sys.argv[3:7])
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    q, k, v = (torch.randn(B, H, S, hd, generator=g).to("cuda", dt) for _ in range(3))
    fn = lambda: F.scaled_dot_product_attention(q, k, v)  # noqa: E731
    flop = 4.0 * B * H * S * S * hd
else:
    M, N, K = (int(v) for v in sys.argv[2:5])
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    a = torch.randn(M, K, generator=g).to("cuda", dt)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", dt)
    b = torch.randn(N, generator=g).to("cuda", dt)
    fn = lambda: F.linear(a, w, b)  # noqa: E731
    flop = 2.0 * M * N * K
for _ in range(3):
    fn()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(iters):
    fn()
torch.cuda.synchronize()
us = (time.perf_counter() - t) / iters * 1e6
print(f"{' '.join(sys.argv[1:])}: {us:.1f} us per call (wall, {iters} calls)  {flop / us / 1e6:.1f} TFLOP/s")
