"""Developer probe (GPU box): which host-side ops of a training step issue device-to-device memcpys (torch profiler)."""
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
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(3):
    tr.step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(images, targets)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name in ("aten::copy_", "aten::clone", "aten::_to_copy", "aten::contiguous", "aten::fill_", "aten::zero_")]
import collections
agg = collections.Counter()
for e in ev:
    st = [f for f in (e.stack or []) if "sihl_amd" in f or "bench.py" in f or "torch/optim" in f or "autograd" in f][:2]
    agg[(e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1][:60] for s in st))] += 1
for k, v in agg.most_common(40):
    print(v, k)
