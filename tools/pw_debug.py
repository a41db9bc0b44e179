"""Developer check (GPU box): which of the two pointwise kernels deviates, where, and how often (fresh buffers each round)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import _C, ops
lib = _C.lib()
torch.manual_seed(0)
for (N, H, W, Cin, Cout) in [(4, 64, 64, 64, 256), (8, 32, 32, 128, 512), (1, 1, 21824, 256, 256)]:
    bad = {0: 0, 2: 0, 6: 0}
    for rnd in range(12):
        x = torch.randn(N, H, W, Cin, device="cuda", dtype=torch.bfloat16)
        w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * Cin ** -0.5).bfloat16()
        b = torch.randn(Cout, device="cuda")
        ref = (x.float().reshape(-1, Cin) @ w.float().reshape(Cout, Cin).T + b).bfloat16()
        for mode in (0, 2, 6):
            lib.sihl_conv2d_pw_enable(mode)
            y, _ = ops.conv2d_raw(x, w, b, 1, 0, 1)
            torch.cuda.synchronize()
            d = (y.reshape(-1, Cout).float() - ref.float()).abs()
            wrong = d > 0.1
            if wrong.any():
                bad[mode] += 1
                rows = wrong.any(1).nonzero().flatten()
                cols = wrong.any(0).nonzero().flatten()
                print(f"shape {(N,H,W,Cin,Cout)} round {rnd} mode {mode}: {int(wrong.sum())} wrong; rows {rows[:6].tolist()}..{rows[-3:].tolist()} (n={len(rows)}) cols {cols[:6].tolist()}..{cols[-3:].tolist()} (n={len(cols)}) max {float(d.max()):.3f}", flush=True)
    print((N, H, W, Cin, Cout), "rounds with deviations per mode (0 tile, 2 pw exact, 6 pw conservative):", bad, flush=True)
lib.sihl_conv2d_pw_enable(1)
