"""Developer probe (GPU box): the north-star sub-metric - BiFPN(3-7,256,3 layers) + ObjectDetection.forward at
bs=32, 512^2, eval mode, bf16 and fp32 - with HIP-event timing."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sihl_amd  # noqa: E402

CH = [3, 64, 256, 512, 1024, 2048]
dev = "cuda"
torch.manual_seed(0)
neck = sihl_amd.layers.BiFPN(CH, 256, 3, 7).to(dev).to(memory_format=torch.channels_last).eval()
head = sihl_amd.heads.ObjectDetection(neck.out_channels, 80, 3, 7).to(dev).to(memory_format=torch.channels_last).eval()
from sihl_amd import ops  # noqa: E402
prep = ops.PreparedWeights(torch.nn.ModuleList([neck, head]))  # inference: bf16 operand copies made once
for dt in (torch.bfloat16, torch.float32):
    g = torch.Generator(device=dev).manual_seed(1)
    levels = [torch.zeros(32, 3, 512, 512, device=dev)] + [
        torch.randn(32, c, 512 // 2 ** l, 512 // 2 ** l, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
        for l, c in enumerate(CH) if l > 0]
    with torch.no_grad():
        for _ in range(3):
            out = head(neck(levels))
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 10
        e[0].record()
        for _ in range(n):
            feats = neck(levels)
        e[1].record()
        for _ in range(n):
            out = head(feats)
        e[2].record()
        torch.cuda.synchronize()
    tn, th = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
    gflop = 32 * (45.64 + 3.69)
    print(f"{str(dt)[6:]:9s} neck {tn:.3f} ms  head.forward {th:.3f} ms  total {tn + th:.3f} ms  -> "
          f"{32 / (tn + th) * 1e3:.0f} img/s, {gflop / (tn + th):.0f} TFLOP/s algorithmic "
          f"({gflop / (tn + th) / (2500 if dt == torch.bfloat16 else 157.3) * 100:.1f} % of the dense MFMA peak)", flush=True)

# ---- the same forward captured into a HIP graph (static shapes, no host syncs in eval forward)
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
levels = [torch.zeros(32, 3, 512, 512, device=dev)] + [
    torch.randn(32, c, 512 // 2 ** l, 512 // 2 ** l, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    for l, c in enumerate(CH) if l > 0]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(3):
        out = head(neck(levels))
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(graph):
    g_out = head(neck(levels))
torch.cuda.synchronize()
with torch.no_grad():
    ref = head(neck(levels))
graph.replay()
torch.cuda.synchronize()
for a, b in zip(g_out, ref):
    assert torch.equal(a, b), "graph replay differs from eager"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    graph.replay()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"bf16 HIP-graph replay: {t:.3f} ms -> {32 / t * 1e3:.0f} img/s, {32 * 49.33 / t:.0f} TFLOP/s algorithmic ({32 * 49.33 / t / 25:.1f} % of MFMA peak)")
