"""Developer probe (GPU box): the 3x3 convs of the small pyramid levels - conv_small.hip against the general kernel
(sihl_conv2d_small_enable 1 / 0), inference epilogue (ReLU + folded BatchNorm) and training epilogue (ReLU + statistics),
GPU time per launch from a HIP-graph replay of 40 launches (no host gaps, no timing events between the launches)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
NB = 8


def timed(fn, n=40):
    for i in range(NB):
        fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(NB):
            fn(i)  # the stream's own workspace / ticket buffers exist before the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(n):
                fn(i % NB)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


for name, N, W in (("L5 16x16", 32, 16), ("L6 8x8", 32, 8), ("L7 4x4", 32, 4), ("L5 bs 8", 8, 16), ("L5 bs 128", 128, 16)):
    xs = [torch.randn(N, W, W, 256, device=dev, dtype=dt) for _ in range(NB)]
    w = torch.randn(256, 3, 3, 256, device=dev, dtype=dt) * 0.02
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    line = f"{name:10s} {2 * N * W * W * 256 * 256 * 9 / 1e9:5.1f} GF"
    for label, kw in (("eval", dict(act="relu", post=(sc, sh))), ("train", dict(act="relu", stats_mode=2))):
        for on in (0, 1):
            lib.sihl_conv2d_small_enable(on)
            t = timed(lambda i: ops.conv2d_raw(xs[i], w, None, 1, 1, 1, **kw))
            line += f" | {label} {'small' if on else 'general'} {t:6.1f} us"
    lib.sihl_conv2d_small_enable(1)
    print(line, flush=True)
    if __import__("os").environ.get("SIHL_HIP_LIB"):  # `make TUNING=1` library: phase ablations (results invalid)
        line = "            ablations (eval):"
        for label, mode in (("all", 0), ("no DMA", 1), ("no MFMA", 2), ("neither", 3), ("no epilogue", 32), ("front only: no DMA, MFMA, epilogue", 35)):
            lib.sihl_conv2d_debug(mode)
            t = timed(lambda i: ops.conv2d_raw(xs[i], w, None, 1, 1, 1, act="relu", post=(sc, sh)))
            line += f" {label} {t:5.1f} |"
        lib.sihl_conv2d_debug(0)
        print(line, flush=True)
