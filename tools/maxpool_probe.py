"""Developer probe (GPU box): stem max-pool forward + backward, sihl kernels vs ATen (bs 32, 256^2 x 64, bf16, NHWC)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import ops  # noqa: E402

dev = "cuda"
x = torch.randn(32, 64, 256, 256, device=dev, dtype=torch.bfloat16).relu().contiguous(memory_format=torch.channels_last)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


xa = x.clone().requires_grad_(True)
ya = F.max_pool2d(xa, 3, 2, 1)
dy = torch.randn_like(ya)
xs = x.permute(0, 2, 3, 1).contiguous().requires_grad_(True)
ys = ops.maxpool3x3s2(xs)
dys = dy.permute(0, 2, 3, 1).contiguous()
print(f"ATen fwd {timeit(lambda: F.max_pool2d(xa, 3, 2, 1)):7.1f} us   bwd {timeit(lambda: torch.autograd.grad(ya, xa, dy, retain_graph=True)):7.1f} us")
print(f"sihl fwd {timeit(lambda: ops.maxpool3x3s2(xs)):7.1f} us   bwd {timeit(lambda: torch.autograd.grad(ys, xs, dys, retain_graph=True)):7.1f} us")
