"""Developer probe (GPU box): the split-K slabs of one training step's weight-gradient launches - bytes written by the wgrad
kernels and read back by the finishing reduction - per layer shape, from the planner's own answers (sihl_conv2d_wgrad_ws_bytes)
to the calls the step makes."""
import collections
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd import ops  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(2):
    tr.step(images, targets)
torch.cuda.synchronize()
calls = collections.Counter()
real = ops._sized


def spy(name, *args):
    v = real(name, *args)
    if name == "sihl_conv2d_wgrad_ws_bytes":
        calls[(args, v)] += 1
    return v


ops._sized = spy
tr.step(images, targets)
torch.cuda.synchronize()
ops._sized = real
rows = []
for (args, ws), cnt in calls.items():
    N, H, W, Cin, Cout, KH, KW, stride, pad, dil, _dt, target = args
    dw = Cout * KH * KW * Cin * 4
    rows.append((ws * cnt, cnt, ws, dw, args))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
print(f"{sum(r[1] for r in rows)} wgrad launches/step; slabs {total / 1e9:.2f} GB written + the same read back = {2 * total / 1e9:.2f} GB/step "
      f"(the gradients themselves: {sum(r[1] * r[3] for r in rows) / 1e9:.3f} GB)")
for tot, cnt, ws, dw, a in rows[:14]:
    print(f"  {tot / 1e6:8.1f} MB  x{cnt:3d}  slab {ws / 1e6:6.1f} MB = {ws / max(1, dw):5.1f} x dW   N{a[0]} {a[1]}x{a[2]} {a[3]}->{a[4]} k{a[5]} s{a[7]} target {a[11]}")
