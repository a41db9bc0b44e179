"""Developer probe (GPU box): host-side cost per op (tiny tensors, so GPU time is negligible)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sihl_amd  # noqa: E402
from sihl_amd import ops  # noqa: E402

dev = "cuda"
blk = sihl_amd.layers.ConvNormAct(64, 64).to(dev).train()
x = torch.randn(2, 64, 8, 8, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
xn = ops.nhwc(x).requires_grad_(True)


def bench(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


print(f"ConvNormAct train fwd (no grad graph): {bench(lambda: blk.forward_nhwc(xn.detach())):7.1f} us/call host")
def fb():
    y = blk.forward_nhwc(xn)
    y.backward(y.detach())
print(f"ConvNormAct train fwd+bwd:             {bench(fb):7.1f} us/call host")
w = torch.ones(2, device=dev, requires_grad=True)
a = torch.randn(2, 4, 4, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
b = torch.randn(2, 8, 8, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
print(f"fuse_up2 fwd:                          {bench(lambda: ops.fuse_up2(a.detach(), b.detach(), w.detach())):7.1f} us/call host")
def fb2():
    o = ops.fuse_up2(a, b, w)
    o.backward(o.detach())
print(f"fuse_up2 fwd+bwd:                      {bench(fb2):7.1f} us/call host")
print(f"torch.empty:                           {bench(lambda: torch.empty((2, 8, 8, 64), device=dev, dtype=torch.bfloat16)):7.1f} us")
print(f"current_stream().cuda_stream:          {bench(lambda: torch.cuda.current_stream().cuda_stream):7.1f} us")
print(f"raw stream:                            {bench(lambda: torch._C._cuda_getCurrentRawStream(0)):7.1f} us")
wt = blk[0].weight
print(f"weight_khwc cast:                      {bench(lambda: ops.weight_khwc(wt, torch.bfloat16)):7.1f} us")

# a chain: one backward call for many blocks (what a real step looks like)
chain = torch.nn.ModuleList([sihl_amd.layers.ConvNormAct(64, 64) for _ in range(20)]).to(dev).train()
def fchain():
    h = xn
    for m in chain:
        h = m.forward_nhwc(h)
    return h
print(f"chain of 20 blocks fwd:                {bench(lambda: fchain(), 50) / 20:7.1f} us/block host")
def fbchain():
    h = fchain()
    h.backward(h.detach())
print(f"chain of 20 blocks fwd+bwd:            {bench(fbchain, 50) / 20:7.1f} us/block host")
