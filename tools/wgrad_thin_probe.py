"""Developer probe (GPU box): weight gradients of ResNet layer1's thin pointwise convs (bs 32, 128^2 maps, bf16): the LDS-DMA panel
kernel against the register-staged 128 x 128 kernel (SIHL_WGRAD_NO_THIN_DMA=1 in the environment selects the latter)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
for name, N, HW, Cin, Cout in (("layer1 64 -> 256", 32, 128, 64, 256), ("layer1 256 -> 64", 32, 128, 256, 64),
                               ("layer1 64 -> 64", 32, 128, 64, 64), ("layer2 128 -> 512", 32, 64, 128, 512),
                               ("layer2 256 -> 128 (entry, 128^2 in)", 32, 128, 256, 128), ("cls logits 256 -> 80 on P3", 32, 64, 256, 80)):
    x = torch.randn(N, HW, HW, Cin, device=dev, dtype=dt)
    dy = torch.randn(N, HW, HW, Cout, device=dev, dtype=dt)
    ref = None
    for _ in range(5):
        dw = ops.conv2d_wgrad_raw(x, dy, 1, 1, 1, 0, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        dw = ops.conv2d_wgrad_raw(x, dy, 1, 1, 1, 0, 1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    want = torch.einsum("npc,npk->kc", x.view(N, -1, Cin)[:2].float(), dy.view(N, -1, Cout)[:2].float())
    got2 = ops.conv2d_wgrad_raw(x[:2].contiguous(), dy[:2].contiguous(), 1, 1, 1, 0, 1).view(Cout, Cin)
    err = float((got2 - want).abs().max() / want.abs().max())
    print(f"{name:38s} {us:7.1f} us  ({mb:5.0f} MB of activations: {mb / us * 1e-3 * 1e3:5.2f} TB/s)   rel err vs fp32 einsum on 2 images {err:.1e}")
