"""Developer probe (GPU box): the secondary BASELINE configurations as training steps on one MI355X.
  configs[1]: ResNet50 + FPN(3-5) + SemanticSegmentation, bs 16, 3x512x512, fp32
  configs[4]-like: multitask ObjectDetection + SemanticSegmentation on BiFPN, bs 16, 3x640x640, bf16 (ResNet50 trunk:
                   timm's convnext_base is not available offline)"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer  # noqa: E402

dev = torch.device("cuda", 0)


def boxes_targets(batch, size, g):
    classes, boxes = [], []
    for b in range(batch):
        n = int(torch.randint(0, 9, (1,), generator=g))
        xy = torch.rand(n, 2, generator=g) * (size * 0.75)
        wh = 16 + torch.rand(n, 2, generator=g) * (size * 0.25 - 16)
        boxes.append(torch.cat([xy, xy + wh], dim=1).to(dev))
        classes.append(torch.randint(0, 80, (n,), generator=g).to(dev))
    return {"classes": classes, "boxes": boxes}


def run(name, model, images, targets, amp, steps=8):
    import copy
    for mode, graph in (("eager two-stream", False), ("HIP-graph single-stream", True)):
        _run(f"{name} [{mode}]", copy.deepcopy(model), images, targets, amp, steps, graph)


def _run(name, model, images, targets, amp, steps, graph):
    tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=amp, graph=graph)
    for _ in range(4):
        tr.step(images, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = tr.step(images, targets)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {images.shape[0] / dt:.0f} img/s, {dt * 1e3:.1f} ms/step, loss {float(loss):.3f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
# ---- configs[1]
bb = sihl_amd.ResNetBackbone("resnet50", top_level=5)
neck = sihl_amd.layers.FPN(bb.out_channels, 256, 3, 5)
head = sihl_amd.heads.SemanticSegmentation(neck.out_channels, num_classes=21, bottom_level=3, top_level=5)
model = sihl_amd.SihlModel(bb, neck, [head]).to(dev).to(memory_format=torch.channels_last)
images = torch.rand(16, 3, 512, 512, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
target = torch.randint(0, 21, (16, 512, 512), generator=g).to(dev)
run("configs[1] ResNet50+FPN+SemanticSegmentation bs16 512^2 fp32", model, images, [target], None)
del model
torch.cuda.empty_cache()
torch.cuda.reset_peak_memory_stats()
# ---- multitask
bb = sihl_amd.ResNetBackbone("resnet50", top_level=5)
neck = sihl_amd.layers.BiFPN(bb.out_channels, 256, 3, 7)
od = sihl_amd.heads.ObjectDetection(neck.out_channels, num_classes=80, bottom_level=3, top_level=7)
ss = sihl_amd.heads.SemanticSegmentation(neck.out_channels, num_classes=21, bottom_level=3, top_level=5)
model = sihl_amd.SihlModel(bb, neck, [od, ss]).to(dev).to(memory_format=torch.channels_last)
images = torch.rand(16, 3, 640, 640, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
targets = [boxes_targets(16, 640, g), torch.randint(0, 21, (16, 640, 640), generator=g).to(dev)]
run("configs[4]-like ResNet50+BiFPN+{ObjectDetection,SemanticSegmentation} bs16 640^2 bf16", model, images, targets, torch.bfloat16)
del model
torch.cuda.empty_cache()
torch.cuda.reset_peak_memory_stats()
# ---- the reference examples' default neck
bb = sihl_amd.ResNetBackbone("resnet50", top_level=5)
neck = sihl_amd.layers.HybridEncoder(bb.out_channels, 256, 3, 7)
od = sihl_amd.heads.ObjectDetection(neck.out_channels, num_classes=80, bottom_level=3, top_level=7)
model = sihl_amd.SihlModel(bb, neck, [od]).to(dev).to(memory_format=torch.channels_last)
images = torch.rand(32, 3, 512, 512, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
run("ResNet50+HybridEncoder(3-7)+ObjectDetection bs32 512^2 bf16", model, images, [boxes_targets(32, 512, g)], torch.bfloat16)
