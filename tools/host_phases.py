"""Developer probe (GPU box): host-side time of each phase of an eager training step (perf_counter around the phases,
no device syncs inside the loop)."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(3):
    tr.step(images, targets)
torch.cuda.synchronize()
acc = {}


def lap(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + t - t0
    return t


N = 10
for _ in range(N):
    t = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True); t = lap("zero_grad", t)
    loss, metrics = tr.forward_loss(images, targets); t = lap("forward_loss", t)
    tr._backward(loss); t = lap("backward (+ join)", t)
    tr.averager.finish(); t = lap("averager.finish", t)
    tr._clip_gradients(); t = lap("clip gradients", t)
    tr.optimizer.step(); t = lap("optimizer.step", t)
    tr.prepared.refresh(); t = lap("prepared.refresh", t)
    tr._step_scheduler(); t = lap("scheduler", t)
torch.cuda.synchronize()
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:22s} {v / N * 1e3:7.2f} ms/step")
print(f"{'total host':22s} {tot / N * 1e3:7.2f} ms/step")
