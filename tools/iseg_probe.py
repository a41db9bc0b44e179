"""Developer probe (GPU box): InstanceSegmentation.forward on BiFPN outputs at bs 32, 512^2 and the fused mask-decode
kernel alone (output-write bound: B*K*H*W elements)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sihl_amd  # noqa: E402
from sihl_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
CH = [3, 64, 256, 256, 256, 256, 256, 256]
head = sihl_amd.heads.InstanceSegmentation(CH, 80, 3, 3, 7).to(dev).to(memory_format=torch.channels_last).eval()
for dt in (torch.bfloat16, torch.float32):
    levels = [torch.zeros(32, 3, 512, 512, device=dev)] + [
        torch.randn(32, c, 512 // 2 ** l, 512 // 2 ** l, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        for l, c in enumerate(CH) if l > 0]
    feats = torch.randn(32, 64, 64, 8, device=dev).to(dt)
    dyn = torch.randn(3200, 176, device=dev).to(dt)[:, :169]
    idx = torch.randint(0, 5456, (32, 100), device=dev, dtype=torch.int32)
    hw = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    with torch.no_grad():
        for _ in range(3):
            head(levels)
            ops.iseg_mask_decode(feats, dyn, idx, hw, (512, 512))
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 10
        e[0].record()
        for _ in range(n):
            out = head(levels)
        e[1].record()
        for _ in range(n):
            m = ops.iseg_mask_decode(feats, dyn, idx, hw, (512, 512))
        e[2].record()
        torch.cuda.synchronize()
    t_head, t_dec = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
    out_bytes = m.numel() * m.element_size()
    print(f"{str(dt)[6:]:9s} head.forward {t_head:.3f} ms | mask decode alone {t_dec * 1e3:.0f} us for {out_bytes / 1e9:.2f} GB of masks "
          f"= {out_bytes / t_dec / 1e9:.2f} TB/s written", flush=True)
    del levels, out, m
