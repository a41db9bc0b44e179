import os, sys, types, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench, sihl_amd
dev = torch.device("cuda", 0)
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(ns, dev)
from sihl_amd import ops
prep = ops.PreparedWeights(model, torch.bfloat16)
for rnd in range(3):
    for cat in ("", "1"):
        sihl_amd.heads.object_detection.CAT_LATERALS = bool(cat)
        r = bench.north_star_forward(model, dev, torch.bfloat16, 32, 512, iters=40)
        print("cat" if cat else "into-flat", round(r["ms"], 3), flush=True)
