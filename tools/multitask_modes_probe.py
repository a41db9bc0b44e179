"""Developer probe (GPU box): the configs[4]-like multitask step (ResNet50 + BiFPN + {ObjectDetection, SemanticSegmentation}, bs 16,
640^2, bf16) - the loss of every step, and of each head, in the execution modes of the Trainer, from identical initial replicas."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)


def boxes_targets(batch, size):
    classes, boxes = [], []
    for b in range(batch):
        n = int(torch.randint(0, 9, (1,), generator=g))
        xy = torch.rand(n, 2, generator=g) * (size * 0.75)
        wh = 16 + torch.rand(n, 2, generator=g) * (size * 0.25 - 16)
        boxes.append(torch.cat([xy, xy + wh], dim=1).to(dev))
        classes.append(torch.randint(0, 80, (n,), generator=g).to(dev))
    return {"classes": classes, "boxes": boxes}


bb = sihl_amd.ResNetBackbone("resnet50", top_level=5)
neck = sihl_amd.layers.BiFPN(bb.out_channels, 256, 3, 7)
od = sihl_amd.heads.ObjectDetection(neck.out_channels, num_classes=80, bottom_level=3, top_level=7)
ss = sihl_amd.heads.SemanticSegmentation(neck.out_channels, num_classes=21, bottom_level=3, top_level=5)
heads = {"od": [od], "ss": [ss], "both": [od, ss]}[sys.argv[1] if len(sys.argv) > 1 else "both"]
model = sihl_amd.SihlModel(bb, neck, heads).to(dev).to(memory_format=torch.channels_last)
images = torch.rand(16, 3, 640, 640, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
tg = {"od": boxes_targets(16, 640), "ss": torch.randint(0, 21, (16, 640, 640), generator=g).to(dev)}
targets = [tg["od" if h is od else "ss"] for h in heads]
if "torchclip" in sys.argv[2:]:
    from sihl_amd import ops
    ops.grad_clip_supported = lambda grads: False
if "noclip" in sys.argv[2:]:
    CLIP = None
else:
    CLIP = 0.1
for mode, kw in (("eager two-stream", {}), ("eager single-stream", {"wgrad_stream": "off"}), ("eager two-stream again", {}),
                 ("HIP graph", {"graph": True})):
    tr = Trainer(copy.deepcopy(model), lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=CLIP,
                 autocast_dtype=torch.bfloat16, **kw)
    out = []
    for i in range(10):
        loss, m = tr.step(images, targets)
        out.append(float(loss))
    torch.cuda.synchronize()
    print(f"{mode:24s} " + " ".join(f"{v:8.3f}" for v in out), flush=True)
