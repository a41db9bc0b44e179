"""Developer probe (GPU box): which ATen ops launch the small copy / fill kernels of a training step (by op name and
input shapes; single-stream step so that everything is on one timeline)."""
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

if "nofused" in sys.argv[1:]:
    from sihl_amd.heads import object_detection as _od
    _od.FUSED_LOSS = False
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(images, targets)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None:
        kids = sum(1 for k in prof.events() if False)
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name in (
            "aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous",
            "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::_to_copy", "aten::cat", "aten::index_select",
            "aten::mul", "aten::sum", "aten::select_backward", "aten::slice_backward", "aten::index_add_"):
        parent = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
        gp = ev.cpu_parent.cpu_parent.name if ev.cpu_parent is not None and ev.cpu_parent.cpu_parent is not None else "-"
        by[(ev.name, parent[:40], gp[:40], str(ev.input_shapes)[:80])] += 1
for (n, par, gp, shp), c in by.most_common(400 if "all" in sys.argv[1:] else 60):
    print(f"{c:4d}  {n:18s} <- {par:40s} <- {gp:40s} {shp}")
