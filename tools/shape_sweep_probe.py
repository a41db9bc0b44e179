"""Developer probe (GPU box): the north-star model at image sizes other than 512^2 (rectangular maps, levels that miss the special
kernels' shapes): training losses and inference outputs with the shape-specialised conv kernels (conv_halo, conv_pyr, conv_small)
against the general tile kernel alone, from identical replicas."""
import copy
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd import _C  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
lib = _C.lib()
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
base = bench.build_model(ns, dev)
worst = 0.0
for (H, W, B) in ((512, 384, 8), (256, 512, 8), (256, 256, 32), (768, 512, 4), (512, 512, 8), (640, 640, 4), (1024, 1024, 2), (512, 1024, 4)):  # multiples of 128 (levels 3-7)
    g = torch.Generator().manual_seed(H * 7 + W)
    images = torch.rand(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    classes, boxes = [], []
    for b in range(B):
        n = int(torch.randint(0, 6, (1,), generator=g))
        xy = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.6, H * 0.6])
        wh = 16 + torch.rand(n, 2, generator=g) * torch.tensor([W * 0.3, H * 0.3])
        boxes.append(torch.cat([xy, xy + wh], dim=1).to(dev))
        classes.append(torch.randint(0, 80, (n,), generator=g).to(dev))
    targets = [{"classes": classes, "boxes": boxes}]
    res = {}
    for mode in ("special", "general"):
        lib.sihl_conv2d_small_enable(1 if mode == "special" else 0)
        lib.sihl_conv2d_halo_enable(1 if mode == "special" else 0)
        model = copy.deepcopy(base)
        tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16)
        losses = [float(tr.step(images, targets)[0]) for _ in range(4)]
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            n_, scores, cls_, bx = model(images)[0] if isinstance(model(images), (list, tuple)) and len(model(images)) == 1 else model.heads[0].forward(model.extract_features(images))
        res[mode] = (losses, scores.float().clone(), bx.float().clone())
        torch.cuda.synchronize()
    lib.sihl_conv2d_small_enable(1)
    lib.sihl_conv2d_halo_enable(1)
    la, lb = res["special"][0], res["general"][0]
    rel = max(abs(a - b) / max(1e-6, abs(b)) for a, b in zip(la, lb))
    ds = float((res["special"][1] - res["general"][1]).abs().max())
    worst = max(worst, rel)
    print(f"{H:4d} x {W:4d} bs {B:2d}: losses special {' '.join(f'{v:8.3f}' for v in la)} | general {' '.join(f'{v:8.3f}' for v in lb)} | max rel gap {rel:.1e}; "
          f"inference scores max abs gap after training {ds:.2e}", flush=True)
print(f"largest relative loss gap: {worst:.1e}")
