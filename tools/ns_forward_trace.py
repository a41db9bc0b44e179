"""North-star forward under rocprofv3 (GPU box): BiFPN(3-7, 256, 3 layers) + ObjectDetection.forward, eval, bs 32,
512^2, bf16 - N forwards, each preceded by a tiny marker launch (`torch.cuda._sleep`-free: a 1-element fill) so that
`profiles/summarize_ns_forward.py` can cut the kernel trace into forwards.

    rocprofv3 --kernel-trace -d gpurun_out/ns_fwd -o ns -- python3 tools/ns_forward_trace.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sihl_amd  # noqa: E402
from sihl_amd import ops  # noqa: E402

CH = [3, 64, 256, 512, 1024, 2048]
dev = "cuda"
torch.manual_seed(0)
neck = sihl_amd.layers.BiFPN(CH, 256, 3, 7).to(dev).to(memory_format=torch.channels_last).eval()
head = sihl_amd.heads.ObjectDetection(neck.out_channels, 80, 3, 7).to(dev).to(memory_format=torch.channels_last).eval()
prep = ops.PreparedWeights(torch.nn.ModuleList([neck, head]))
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
levels = [torch.zeros(32, 3, 512, 512, device=dev)] + [
    torch.randn(32, c, 512 // 2 ** l, 512 // 2 ** l, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    for l, c in enumerate(CH) if l > 0]
marker = torch.zeros(7, device=dev, dtype=torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
if os.environ.get("SIHL_MLP_STAGES"):
    from sihl_amd import _C  # noqa: E402
    _C.lib().sihl_mlp_stages(int(os.environ["SIHL_MLP_STAGES"]))
if os.environ.get("SIHL_LAYERED_MLP"):
    from sihl_amd.heads import mlp as _mlp  # noqa: E402
    _mlp.FUSE_WHOLE_MLP = False
with torch.no_grad():
    for _ in range(3):
        out = head(neck(levels))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        marker.fill_(1.0)          # the cut mark: the only float64 fill of 7 elements in the trace
        out = head(neck(levels))
    marker.fill_(1.0)
    e1.record()
    torch.cuda.synchronize()
print(f"bf16 north-star forward under trace: {e0.elapsed_time(e1) / n:.3f} ms per forward ({n} forwards)")
