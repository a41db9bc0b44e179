"""Developer probe (GPU box): which Python call sites launch the fill / copy kernels of a training step."""
import collections
import os
import sys
import types

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16,
             wgrad_stream="off")
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(3):
    tr.step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(images, targets)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::clone",
             "aten::contiguous", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "sihl_amd" in s or "bench.py" in s or "torch/optim" in s or "clip_grad" in s
              or "autograd" in s]
        by[(n, tuple(st[:3]))] += 1
for (n, st), c in by.most_common(40):
    print(f"{c:5d}  {n:18s}  {' <- '.join(s.split('/')[-1][:70] for s in st)}")
