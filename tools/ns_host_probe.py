"""Developer probe (GPU box): is the eager north-star forward (BiFPN + ObjectDetection.forward, eval, bs 32, 512^2, bf16)
bound by the GPU or by the host issuing its ~90 launches?  Host issue time per forward (no sync inside), wall per forward,
and the same forward as a HIP-graph replay."""
import sys
import time
import types

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev).eval()
prep = ops.PreparedWeights(model, torch.bfloat16)
chans = [3, 64, 256, 512, 1024, 2048]
g = torch.Generator(device=dev).manual_seed(1)
levels = [torch.zeros(32, 3, 512, 512, device=dev)] + [
    torch.randn(32, c, 512 // 2 ** l, 512 // 2 ** l, device=dev, generator=g).to(torch.bfloat16)
    .contiguous(memory_format=torch.channels_last) for l, c in enumerate(chans) if l > 0]
fwd = lambda: model.heads[0](model.neck(levels))  # noqa: E731
with torch.no_grad():
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    for rnd in range(3):
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            fwd()
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n
        print(f"eager: host issue {host * 1e3:.3f} ms / forward, wall {wall * 1e3:.3f} ms / forward", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fwd()
    torch.cuda.synchronize()
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"HIP-graph replay: {e0.elapsed_time(e1) / 40:.3f} ms / forward", flush=True)
