"""Developer probe (GPU box): the P3 3x3 conv (bs 32, 64x64, 256 -> 256, bf16) on the general 256 x 256 tile against
conv_halo.hip (sihl_conv2d_halo_enable 0 / 1), eval and training epilogues, forward and dgrad; GPU time per launch from a
HIP-graph replay of 20 launches over rotating operands."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
lib = _C.lib()
NB = 6


def timed(fn, n=20):
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


VARIANTS = ((0, "general tile"), (3, "halo, barrier in front of the stage"), (2, "halo, barrier before the last tap's multiplies"))
for name, N, H, Wd in (("P3 bs 32", 32, 64, 64), ("P4 bs 32", 32, 32, 32), ("P4 bs 64", 64, 32, 32)):
    xs = [torch.randn(N, H, Wd, 256, device=dev, dtype=dt) for _ in range(NB)]
    ws = [torch.randn(256, 3, 3, 256, device=dev, dtype=dt) * 0.02 for _ in range(NB)]
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    gf = 2 * N * H * Wd * 256 * 256 * 9 / 1e9
    for label, kw in (("eval", dict(act="relu", post=(sc, sh))), ("train", dict(act="relu", stats_mode=2))):
        res = {m: [] for m, _ in VARIANTS}
        for rnd in range(5):  # variants interleaved: the clock the chip holds drifts over a run
            for mode, _ in VARIANTS:
                lib.sihl_conv2d_halo_enable(mode)
                res[mode].append(timed(lambda i: ops.conv2d_raw(xs[i], ws[i], None, 1, 1, 1, **kw)))
        lib.sihl_conv2d_halo_enable(1)
        print(f"{name} {label}: " + " | ".join(f"{mname} {sorted(res[m])[2]:6.1f}" for m, mname in VARIANTS) + "  us (median of 5)", flush=True)
