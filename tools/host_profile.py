"""Developer probe (GPU box): where the HOST time of an eager training step goes (cProfile, no device syncs inside)."""
import cProfile
import pstats
import sys
import time
import types

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
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
t0 = time.perf_counter()
for _ in range(5):
    tr.step(images, targets)
host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5
print(f"host-side issue time {host * 1e3:.1f} ms/step, wall {wall * 1e3:.1f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(images, targets)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
