"""Developer probe (GPU box): the 3x3 convs of the small pyramid levels - general tile kernel / conv_small.hip / conv_pyr.hip
(sihl_conv2d_small_enable 0 / 2 / 1) - and a BiFPN node as [fusion kernel -> conv] against the one-launch form
(ops.pyr_conv_raw with a fused producer); GPU time per launch from a HIP-graph replay of 40 launches."""
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
            fn(i)
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
        for mode, mname in ((0, "general"), (2, "small"), (1, "pyr")):
            lib.sihl_conv2d_small_enable(mode)
            if mode == 1 and kw.get("stats_mode") and W != 16:
                t = timed(lambda i: ops.pyr_conv_raw(w, x=xs[i], **kw))
            else:
                t = timed(lambda i: ops.conv2d_raw(xs[i], w, None, 1, 1, 1, **kw))
            line += f" | {label} {mname} {t:5.1f}"
    lib.sihl_conv2d_small_enable(1)
    print(line + " us", flush=True)

print("BiFPN nodes, eval epilogue: [fusion kernel -> conv (pyr)] vs one launch", flush=True)
for name, N, W in (("P5", 32, 16), ("P6", 32, 8), ("P7", 32, 4)):
    w = torch.randn(256, 3, 3, 256, device=dev, dtype=dt) * 0.02
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    lo = [torch.randn(N, W // 2, W // 2, 256, device=dev, dtype=dt) for _ in range(NB)]
    hi = [torch.randn(N, 2 * W, 2 * W, 256, device=dev, dtype=dt) for _ in range(NB)]
    b = [torch.randn(N, W, W, 256, device=dev, dtype=dt) for _ in range(NB)]
    c = [torch.randn(N, W, W, 256, device=dev, dtype=dt) for _ in range(NB)]
    w2, w3 = torch.randn(2, device=dev), torch.randn(3, device=dev)
    kw = dict(act="relu", post=(sc, sh))
    with torch.no_grad():
        t_sep = timed(lambda i: ops.pyr_conv_raw(w, x=ops.fuse_up2(lo[i], b[i], w2), **kw))
        t_one = timed(lambda i: ops.pyr_conv_raw(w, fuse=("up2", lo[i], b[i], w2), **kw)) if W < 16 else float("nan")
        line = f"{name}: up2 node {t_sep:5.1f} -> {t_one:5.1f} us"
        if W < 16:
            t_sep = timed(lambda i: ops.pyr_conv_raw(w, x=ops.blur_fuse(hi[i], b[i], c[i], w3), **kw))
            t_one = timed(lambda i: ops.pyr_conv_raw(w, fuse=("blur", hi[i], b[i], c[i], w3, None), **kw))
            line += f" | blur node {t_sep:5.1f} -> {t_one:5.1f} us"
    print(line, flush=True)
