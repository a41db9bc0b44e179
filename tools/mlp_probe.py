"""Developer probe (GPU box): the whole-MLP kernel (sihl_mlp_fwd) on the detection head's loc MLP - 174 592 rows x 256,
4 hidden layers, 1 output - against the layer-by-layer kernels; with a `make TUNING=1` library (SIHL_HIP_LIB) also the
phase ablations of MlpParams::dbg."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sihl_amd import _C, ops  # noqa: E402
from sihl_amd.heads import mlp as mlp_mod  # noqa: E402

dev = "cuda"
lib = _C.lib()
torch.manual_seed(0)


def timed(fn, n=20):
    """us per call of fn, GPU time: n calls captured into a HIP graph (no host gaps between the launches), 5 replays."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(n):
                fn()
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


for rows, cout in ((174592, 1), (3200, 80)):
    m = mlp_mod.MLP(256, [256] * 4 + [cout], norm_layer=torch.nn.LayerNorm, activation_layer=torch.nn.SiLU).to(dev).eval()
    prep = ops.PreparedWeights(m)
    x = torch.randn(rows, 256, device=dev).bfloat16()
    with torch.no_grad():
        mlp_mod.FUSE_WHOLE_MLP = False
        t_lay = timed(lambda: m(x))
        mlp_mod.FUSE_WHOLE_MLP = True
        ops.MLP_KERNEL = "rows"
        t_rows = timed(lambda: m(x))
        ops.MLP_KERNEL = "tile"
        line = f"rows {rows} -> {cout}: layered {t_lay:7.1f} us | registers (mlp_rows) {t_rows:7.1f} us"
        for st in (2, 3):
            lib.sihl_mlp_stages(st)
            line += f" | one launch, {st} stages {timed(lambda: m(x)):7.1f} us"
        print(line, flush=True)
        # the kernel alone: direct C-ABI launches (the Python wrapper costs ~35 us per call, more than a small launch)
        lin = [q for q in m if isinstance(q, torch.nn.Linear)]
        lns = [q for q in m if isinstance(q, torch.nn.LayerNorm)]
        plan = ops._mlp_plan(lin, lns, torch.bfloat16)
        cp = (cout + 7) // 8 * 8
        out = torch.empty(rows, cp, device=dev, dtype=torch.bfloat16)

        def direct():
            lib.sihl_mlp_fwd(x.data_ptr(), 256, rows, 256, 256, 4, plan.w, plan.bias, plan.gamma, plan.beta, 1e-5, 2, cout,
                             out.data_ptr(), cp, 1, torch.cuda.current_stream().cuda_stream)
        ops.MLP_KERNEL = "rows"
        plan_r = ops._mlp_plan(lin, lns, torch.bfloat16)
        ops.MLP_KERNEL = "tile"

        def direct_rows():
            lib.sihl_mlp_rows_fwd(x.data_ptr(), 256, rows, 256, 256, 4, plan_r.w_rows, plan_r.bias, plan_r.gamma, plan_r.beta,
                                  1e-5, 2, cout, out.data_ptr(), cp, 1, torch.cuda.current_stream().cuda_stream)
        print(f"   direct launches, registers kernel: {timed(direct_rows, 50):6.1f} us", flush=True)
        if os.environ.get("SIHL_HIP_LIB"):
            line = "   registers kernel ablations:"
            for name, mode in (("all", 0), ("no LN", 1), ("no DMA", 2), ("no MFMA", 4), ("no LN no DMA", 3), ("no LN no MFMA", 5),
                               ("no DMA no MFMA", 6), ("skeleton", 7)):
                lib.sihl_mlp_rows_debug(mode)
                line += f" {name} {timed(direct_rows, 50):6.1f} |"
            lib.sihl_mlp_rows_debug(0)
            print(line, flush=True)
        m = None
        timed_ = lambda: timed(direct, 50)  # noqa: E731
        line = f"   direct launches:"
        for st in (2, 3):
            lib.sihl_mlp_stages(st)
            line += f" {st} stages {timed_():7.1f} us |"
        print(line, flush=True)
        if os.environ.get("SIHL_HIP_LIB"):
            timed_m = timed
            timed = lambda fn, n=50: timed_m(direct, n)  # noqa: E731
            for st in (2, 3):
                lib.sihl_mlp_stages(st)
                line = f"   ablations, {st} stages:"
                for name, mode in (("all", 0), ("no LN", 1), ("no z write", 2), ("no LN, no z", 3), ("no MFMA", 4), ("no weight DMA", 8),
                                   ("no MFMA no DMA", 12), ("only DMA", 7), ("skeleton", 15)):
                    lib.sihl_mlp_debug(mode)
                    line += f" {name} {timed(lambda: m(x)):6.1f} |"
                lib.sihl_mlp_debug(0)
                print(line, flush=True)
            timed = timed_m
