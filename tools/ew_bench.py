"""Developer probe (GPU box): achieved HBM bandwidth of the streaming kernels vs torch copy/add yardsticks.
Buffers rotate over a >1 GB working set so the 256 MB Infinity Cache cannot serve repeats."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import ops  # noqa: E402

dev = "cuda"
dt = torch.bfloat16


def timeit(fn, nbuf, iters=24):
    for i in range(nbuf):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % nbuf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for rows, C in ((131072, 256), (174592, 256), (524288, 256), (524288, 64), (32768, 1024)):
    nbytes = rows * C * 2
    nbuf = max(2, int(1.5e9 // (3 * nbytes)))
    xs = [torch.randn(rows, C, device=dev).to(dt) for _ in range(nbuf)]
    ds = [torch.randn(rows, C, device=dev).to(dt) for _ in range(nbuf)]
    ys = [torch.empty(rows, C, device=dev, dtype=dt) for _ in range(nbuf)]
    scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, rstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    res = []

    def rep(name, us, passes):
        res.append(f"{name} {us:6.1f}us {passes * nbytes / us / 1e6:5.2f}TB/s")

    rep("copy", timeit(lambda i: ys[i].copy_(xs[i]), nbuf), 2)
    rep("add", timeit(lambda i: torch.add(xs[i], ds[i], out=ys[i]), nbuf), 3)
    rep("affine_act", timeit(lambda i: ops.affine_act(xs[i], scale, shift, "relu"), nbuf), 2)
    rep("affine_add_act", timeit(lambda i: ops.affine_add_act(xs[i], ds[i], scale, shift, "relu"), nbuf), 3)
    rep("affine_bwd", timeit(lambda i: ops.affine_act_bwd(xs[i], ds[i], scale, shift, "relu"), nbuf), 3)
    rep("bn_bwd(m0)", timeit(lambda i: ops.norm_act_bwd(xs[i], ds[i], mean, rstd, g, b, 0, "relu", True), nbuf), 5)
    rep("bn_bwd(m1)", timeit(lambda i: ops.norm_act_bwd(xs[i], ds[i], mean, rstd, g, b, 1, "relu", True), nbuf), 5)
    rep("colsum", timeit(lambda i: ops.colsum(xs[i]), nbuf), 1)
    if C == 256:
        zs = [x.clone().requires_grad_(True) for x in xs]
        gp, bp = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rep("ln_fwd", timeit(lambda i: ops.layernorm_act(xs[i], g, b), nbuf), 2)
        outs = [ops.layernorm_act(z, gp, bp) for z in zs]

        def lnb(i):
            torch.autograd.grad(outs[i], (zs[i], gp, bp), ds[i], retain_graph=True)
        rep("ln_bwd", timeit(lnb, nbuf), 3)
    print(f"rows {rows} C {C} ({nbytes / 1e6:.0f} MB/tensor, {nbuf} sets): " + " | ".join(res), flush=True)
