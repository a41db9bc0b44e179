"""Developer probe (GPU box): same-box A/B of the north-star forward (bench.north_star_forward: BiFPN + ObjectDetection.forward,
eval, bs 32, 512^2, bf16) under the library's switches - boxes of the pool differ by several per cent, so only alternating
runs in one process compare.  Usage: python tools/ns_ab.py [rounds]"""
import sys
import types

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd import _C, ops  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
prep = ops.PreparedWeights(model, torch.bfloat16)
lib = _C.lib()


def variant(small, mlp, waves=4, halo=1):
    lib.sihl_conv2d_halo_enable(halo)
    lib.sihl_conv2d_small_enable(small)
    lib.sihl_mlp_rows_config(waves)
    ops.MLP_KERNEL = mlp


# (round 4: the K-loop rotation / dispatch-rule switches exist in `make TUNING=1` libraries only; their A/Bs are
# profiles/r03_ns_ab.txt)
VARIANTS = [("default (conv_pyr on P5-P7, register MLP)", 1, "rows"),
            ("conv_small on P5-P7 (round-3 kernel)", 2, "rows"),
            ("general conv on P5-P7", 0, "rows"), ("LDS-tile MLP", 1, "tile"),
            ("register MLP as one 8-wave workgroup per CU, 4-stage ring", 1, "rows", 8),
            ("P3 3x3 on the general 256x256 tile (no halo-resident patch)", 1, "rows", 4, 0)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {v[0]: [] for v in VARIANTS}
for rnd in range(rounds):
    for name, small, mlp, *rest in VARIANTS:
        variant(small, mlp, *rest)
        r = bench.north_star_forward(model, dev, torch.bfloat16, 32, 512, iters=40)
        res[name].append(r["ms"])
variant(1, "rows")
for name in res:
    v = res[name]
    print(f"{name:80s} " + " ".join(f"{x:.3f}" for x in v) + f"   median {sorted(v)[len(v) // 2]:.3f} ms", flush=True)
