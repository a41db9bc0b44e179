"""Developer probe (GPU box): sihl conv kernels on ResNet50 bottleneck shapes (bs 32, 512^2 input), bf16."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
# (name, H=W, Cin, Cout, K, count per forward)
SHAPES = [("l1 1x1 256->64", 128, 256, 64, 1, 2), ("l1 1x1 64->64", 128, 64, 64, 1, 1), ("l1 3x3 64->64", 128, 64, 64, 3, 3),
          ("l1 1x1 64->256", 128, 64, 256, 1, 4),
          ("l2 1x1 512->128", 64, 512, 128, 1, 3), ("l2 3x3 128->128", 64, 128, 128, 3, 4), ("l2 1x1 128->512", 64, 128, 512, 1, 4),
          ("l3 1x1 1024->256", 32, 1024, 256, 1, 5), ("l3 3x3 256->256", 32, 256, 256, 3, 6), ("l3 1x1 256->1024", 32, 256, 1024, 1, 6),
          ("l4 1x1 2048->512", 16, 2048, 512, 1, 2), ("l4 3x3 512->512", 16, 512, 512, 3, 3), ("l4 1x1 512->2048", 16, 512, 2048, 1, 3)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


tot_f = tot_w = 0.0
for name, S, Cin, Cout, K, cnt in SHAPES:
    x = torch.randn(32, S, S, Cin, device=dev, dtype=dt)
    w = torch.randn(Cout, K, K, Cin, device=dev, dtype=dt) * 0.05
    dy = torch.randn(32, S, S, Cout, device=dev, dtype=dt)
    flops = 2.0 * 32 * S * S * Cin * Cout * K * K
    tf = timeit(lambda: ops.conv2d_raw(x, w, None, 1, K // 2, 1, stats_mode=1))
    tw = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, K, K, 1, K // 2, 1))
    byts = (x.numel() + dy.numel()) * 2
    print(f"{name:18s} x{cnt} {flops/1e9:7.1f} GF | fwd {tf*1e6:7.1f} us {flops/tf/1e12:6.1f} TF/s {byts/tf/1e12:5.2f} TB/s | wgrad {tw*1e6:7.1f} us {flops/tw/1e12:6.1f} TF/s", flush=True)
    tot_f += tf * cnt
    tot_w += tw * cnt
print(f"projected conv time per step: fwd {tot_f*1e3:.2f} ms + dgrad ~{tot_f*1e3:.2f} ms + wgrad {tot_w*1e3:.2f} ms = {(2*tot_f+tot_w)*1e3:.2f} ms")
