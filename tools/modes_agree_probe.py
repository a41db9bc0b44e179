"""Developer probe (GPU box): do the Trainer's execution modes agree?  For one model per head family (ResNet18 trunk + BiFPN, bs 8,
256^2, bf16, odd class / keypoint counts so that padded operands occur): 8 steps from identical replicas as eager two-stream (twice),
eager single-stream and HIP graph.  Two-stream must reproduce itself bit for bit and follow the single-stream trajectory."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
B, S = 8, 256
g = torch.Generator().manual_seed(3)
images = torch.rand(B, 3, S, S, generator=g).to(dev).contiguous(memory_format=torch.channels_last)


def boxes(n_max=5):
    cls, bx = [], []
    for b in range(B):
        n = int(torch.randint(1, n_max, (1,), generator=g))  # (>= 1: the reference's instance head cannot take an image without instances)
        xy = torch.rand(n, 2, generator=g) * (S * 0.6)
        wh = 16 + torch.rand(n, 2, generator=g) * (S * 0.3 - 16)
        bx.append(torch.cat([xy, xy + wh], dim=1).to(dev))
        cls.append(torch.randint(0, 5, (n,), generator=g).to(dev))
    return cls, bx


def cases(ch):
    H = sihl_amd.heads
    cls, bx = boxes()
    yield "ObjectDetection (5 classes)", H.ObjectDetection(ch, num_classes=5, bottom_level=3, top_level=7), {"classes": cls, "boxes": bx}
    yield "SemanticSegmentation (21)", H.SemanticSegmentation(ch, num_classes=21, bottom_level=3, top_level=5), torch.randint(0, 21, (B, S, S), generator=g).to(dev)
    masks = []
    for b, bb in enumerate(bx):
        m = torch.zeros(bb.shape[0], S, S, dtype=torch.bool)
        for k, (x0, y0, x1, y1) in enumerate(bb.tolist()):
            m[k, int(y0):int(y1), int(x0):int(x1)] = True
        masks.append(m.to(dev))
    yield "InstanceSegmentation (5)", H.InstanceSegmentation(ch, num_classes=5), {"classes": cls, "masks": masks}
    kp = [torch.rand(bb.shape[0], 7, 2, generator=g).to(dev) * S for bb in bx]
    pr = [torch.rand(bb.shape[0], 7, generator=g).to(dev) > 0.2 for bb in bx]
    yield "KeypointDetection (7 keypoints)", H.KeypointDetection(ch, num_keypoints=7, bottom_level=5, top_level=7), {"presence": pr, "keypoints": kp}
    quads = [torch.stack([bb[:, [0, 1]], bb[:, [2, 1]], bb[:, [2, 3]], bb[:, [0, 3]]], dim=1) for bb in bx]
    yield "QuadrilateralDetection (5)", H.QuadrilateralDetection(ch, num_classes=5, bottom_level=3, top_level=7), {"classes": cls, "quads": quads}
    yield "DepthEstimation", H.DepthEstimation(ch, lower_bound=0.1, upper_bound=10.0), {"targets": torch.rand(B, S, S, generator=g).to(dev) * 9 + 0.5, "masks": torch.rand(B, S, S, generator=g).to(dev) > 0.1}


torch.manual_seed(0)
bb = sihl_amd.ResNetBackbone("resnet18", top_level=5)
neck = sihl_amd.layers.BiFPN(bb.out_channels, 64, 3, 7)
worst = 0.0
for name, head, target in cases(neck.out_channels):
    model = sihl_amd.SihlModel(copy.deepcopy(bb), copy.deepcopy(neck), [head]).to(dev).to(memory_format=torch.channels_last)
    rows = {}
    for mode, kw in (("two-stream", {}), ("two-stream again", {}), ("single-stream", {"wgrad_stream": "off"}), ("graph", {"graph": True})):
        try:
            tr = Trainer(copy.deepcopy(model), lr=3e-4, weight_decay=1e-4, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16, **kw)
            rows[mode] = [float(tr.step(images, [target])[0]) for _ in range(8)]
        except Exception as e:  # noqa: BLE001
            rows[mode] = f"{type(e).__name__}: {str(e)[:90]}"
    torch.cuda.synchronize()
    print(name)
    for mode, r in rows.items():
        print(f"   {mode:18s} " + (" ".join(f"{v:9.4f}" for v in r) if isinstance(r, list) else r))
    if all(isinstance(r, list) for r in rows.values()):
        same = rows["two-stream"] == rows["two-stream again"]
        dev_rel = max(abs(a - b) / max(1e-6, abs(b)) for a, b in zip(rows["two-stream"], rows["single-stream"]))
        worst = max(worst, dev_rel)
        print(f"   two-stream reproduces itself: {same}; largest relative gap to single-stream {dev_rel:.2e}; graph == single-stream: {rows['graph'] == rows['single-stream']}")
print(f"largest two-stream / single-stream gap over all heads: {worst:.2e}")
