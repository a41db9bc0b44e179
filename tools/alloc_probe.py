"""Developer probe (GPU box): does the training step reach a steady state in the caching allocator?  Prints, per step,
hipMalloc / hipFree counts and reserved / allocated bytes.  python tools/alloc_probe.py [steps] [wgrad_stream]"""
import sys
import types

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
model = bench.build_model(ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16,
             wgrad_stream=mode)
images, targets = bench.synthetic_batch(32, 512, dev, 0)
prev = None
for i in range(steps):
    tr.step(images, targets)
    if i % 4 == 3:
        torch.cuda.synchronize()  # every 4th step: otherwise the host runs ahead as in the benchmark
    st = torch.cuda.memory_stats(dev)
    cur = (st["num_device_alloc"], st["num_device_free"], st["reserved_bytes.all.current"], st["allocated_bytes.all.current"],
           st["num_alloc_retries"], st.get("num_sync_all_streams", 0))
    d = tuple(c - p for c, p in zip(cur, prev)) if prev else cur
    print(f"step {i:2d}: hipMalloc +{d[0]:4d}  hipFree +{d[1]:4d}  reserved {cur[2] / 2**30:7.2f} GiB ({d[2] / 2**20:+9.1f} MiB)  "
          f"allocated {cur[3] / 2**30:6.2f} GiB  retries {cur[4]}", flush=True)
    prev = cur
