"""Developer probe (GPU box): how long does the main stream wait for the wgrad side stream at the join after backward?
(If ~0 the side stream is off the critical path and only main-stream kernels bound the step.)"""
import sys
import types

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
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
pairs = []
orig = ops.join_side_stream


def timed_join():
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    side = ops.side_stream_in_use()
    a.record()                       # main stream: everything of backward issued so far
    if side is not None:
        c.record(side)               # side stream: behind its last weight gradient
    orig()
    b.record()                       # main stream: after the wait
    pairs.append((a, b, c if side is not None else None))


ops.join_side_stream = timed_join
for _ in range(5):
    tr.step(images, targets)
torch.cuda.synchronize()
pairs.clear()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    tr.step(images, targets)
t1.record()
torch.cuda.synchronize()
print(f"step {t0.elapsed_time(t1) / 10:.2f} ms; joins per step {len(pairs) / 10:.1f}")
waits = [a.elapsed_time(b) for a, b, c in pairs]
behind = [a.elapsed_time(c) for a, b, c in pairs if c is not None]
print("main-stream wait at the join(s), ms per step:", round(sum(waits) / 10, 3), " per join:", [round(w, 3) for w in waits[:6]])
print("side stream's last kernel finishes this long after the main stream reached the join, ms:", [round(w, 3) for w in behind[:6]])
