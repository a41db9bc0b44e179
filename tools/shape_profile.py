"""Developer probe (GPU box): per-shape timing of the two matrix-core kernels inside the real training step.
Launch records are grouped by (algorithmic flops, bytes), which identifies the layer shape."""
import collections
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd import _C  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402
import types  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
for _ in range(3):
    tr.step(images, targets)
lib = _C.lib()
torch.cuda.synchronize()
lib.sihl_profile_enable(1)
STEPS = 3
for _ in range(STEPS):
    tr.step(images, targets)
torch.cuda.synchronize()
lib.sihl_profile_enable(0)
for slot, name in ((0, "conv fwd/dgrad/linear"), (1, "wgrad")):
    n = lib.sihl_profile_records(slot, _C.BF16, None, 0)
    buf = (ctypes.c_double * (3 * n))()
    lib.sihl_profile_records(slot, _C.BF16, buf, n)
    groups = collections.defaultdict(list)
    for i in range(n):
        groups[(buf[3 * i + 1], buf[3 * i + 2])].append(buf[3 * i])
    tot = sum(sum(v) for v in groups.values()) / STEPS
    print(f"== {name}: {n / STEPS:.0f} launches/step, {tot:.2f} ms/step")
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    for (fl, by), ts in rows[:40]:
        avg = sum(ts) / len(ts)
        print(f"  {sum(ts) / STEPS:7.3f} ms/step  x{len(ts) / STEPS:4.0f}  avg {avg * 1e3:7.1f} us  {fl / 1e9:8.2f} GFLOP  {by / 1e6:7.1f} MB  "
              f"{fl / avg / 1e9:7.1f} TFLOP/s  {by / avg / 1e9 * 1e3 / 1e3:6.2f} TB/s-alg  AI {fl / by:6.0f}")
