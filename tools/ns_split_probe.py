"""Developer probe (GPU box): the north-star forward (BiFPN(3-7) + ObjectDetection.forward, eval, bs 32, 512^2, bf16)
as ONE chain against the same forward cut into batch slices that run on separate HIP streams - the dependent chain of
tiny pyramid levels (P5-P7: 3 % of the flops, ~20 % of the time, launches that leave most CUs idle) of one slice can then
run under the other slice's P3 / P4 convs.  Eval mode is per-image (BatchNorm on running statistics), so the slices
compute exactly what the whole batch does.  Eager and as a HIP-graph replay (no host issue time)."""
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
B = 32
levels = [torch.zeros(B, 3, 512, 512, device=dev)] + [
    torch.randn(B, c, 512 // 2 ** l, 512 // 2 ** l, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last)
    for l, c in enumerate(CH) if l > 0]
streams = [torch.cuda.Stream() for _ in range(4)]


def whole():
    return head(neck(levels))


def sliced(nslices, nstreams):
    def run():
        main = torch.cuda.current_stream()
        outs = [None] * nslices
        step = B // nslices
        for s in streams[:nstreams]:
            s.wait_stream(main)
        for k in range(nslices):
            s = streams[k % nstreams]
            with torch.cuda.stream(s):
                lv = [t[k * step:(k + 1) * step] for t in levels]
                outs[k] = head(neck(lv))
        for s in streams[:nstreams]:
            main.wait_stream(s)
        return [torch.cat([o[i] for o in outs]) for i in range(4)]
    return run


def time_eager(fn, n=20):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def time_graph(fn, n=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        out = fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


with torch.no_grad():
    ref = whole()
torch.cuda.synchronize()
print(f"whole batch, one stream: eager {time_eager(whole):.3f} ms", flush=True)
t, _ = time_graph(whole)
print(f"whole batch, one stream: graph {t:.3f} ms", flush=True)
for nsl, nst in ((2, 2), (4, 2), (4, 4), (2, 1)):
    fn = sliced(nsl, nst)
    te = time_eager(fn)
    try:
        tg, out = time_graph(fn)
    except Exception as ex:  # noqa: BLE001
        tg, out = float("nan"), None
        print("  graph capture failed:", type(ex).__name__, str(ex)[:200])
    same = ""
    if out is not None:
        same = " | " + ", ".join(f"{(a.float() - b.float()).abs().max().item():.3g}" for a, b in zip(out, ref))
    print(f"{nsl} slices on {nst} streams: eager {te:.3f} ms | graph {tg:.3f} ms{same}", flush=True)
