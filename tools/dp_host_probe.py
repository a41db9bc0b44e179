"""Developer probe (GPU box): the multi-GPU code path on ONE rank (RCCL process group of one rank, gradient buckets forced): host
time per phase of an eager step and the step time, beside the same Trainer without a process group."""
import cProfile
import os
import pstats
import sys
import time
import types

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dp = "nodp" not in sys.argv[1:]
if dp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=dev)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16,
             force_buckets=dp)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(3):
    tr.step(images, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(images, targets)
host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
print(f"{'DP path, one rank' if dp else 'no process group'}: host issue {host * 1e3:.1f} ms/step, wall {wall * 1e3:.1f} ms/step", flush=True)
acc = {}


def lap(name, t):
    n = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + n - t
    return n


for _ in range(10):
    t = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True); t = lap("zero_grad", t)
    loss, _m = tr.forward_loss(images, targets); t = lap("forward_loss", t)
    tr._backward(loss); t = lap("backward (+ hooks, join)", t)
    tr.averager.finish(); t = lap("averager.finish", t)
    tr._clip_gradients(); t = lap("clip", t)
    tr.optimizer.step(); t = lap("optimizer.step", t)
    if tr.prepared is not None:
        tr.prepared.refresh(); t = lap("prepared.refresh", t)
torch.cuda.synchronize()
for k, v in acc.items():
    print(f"   {k:28s} {v / 10 * 1e3:7.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(images, targets)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
if dp:
    dist.destroy_process_group()
