"""Developer probe (GPU box): kernel launches and GPU time of the detection loss (everything between the head's MLP
outputs and their gradients) at the flagship shapes - what a fused loss kernel would replace."""
import os
import sys

import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.heads.box_ops import complete_box_iou_loss  # noqa: E402

dev = torch.device("cuda", 0)
head = sihl_amd.heads.ObjectDetection([3, 64, 256, 256, 256, 256, 256, 256], 80, 3, 7).to(dev)
_, targets = bench.synthetic_batch(32, 512, dev, 0)
tg = targets[0]
level_hw = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
B, P = 32, 5456


def count(fn, name):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(f"{name:34s} {len(ks):4d} launches  {sum(e.device_time for e in ks) / 1e3:7.3f} ms of kernels", flush=True)


count(lambda: head._targets(level_hw, 512, 512, tg["classes"], tg["boxes"], dev), "target matching (prematch)")
t = head._targets(level_hw, 512, 512, tg["classes"], tg["boxes"], dev)
R = t.rows.numel()
g = torch.Generator(device=dev).manual_seed(0)
loc = torch.randn(B, P, device=dev, generator=g, dtype=torch.bfloat16).requires_grad_(True)
iou = torch.randn(B, P, device=dev, generator=g, dtype=torch.bfloat16).requires_grad_(True)
box = (0.1 * torch.randn(R, 4, device=dev, generator=g)).bfloat16().requires_grad_(True)
cls = torch.randn(R, 80, device=dev, generator=g, dtype=torch.bfloat16).requires_grad_(True)


def loss_all():
    loc_loss = F.binary_cross_entropy_with_logits(loc.float(), t.loc_target, reduction="none").sum() / t.loc_norm
    iou_loss = F.mse_loss(iou.float(), t.rel_iou, reduction="none").sum() / t.iou_norm
    box_preds = t.cand_offsets + t.cand_scales * box.float().exp()
    box_loss = (t.wts * complete_box_iou_loss(box_preds, t.tgt_box)).sum() / t.wsum
    cls_loss = (t.wts * F.cross_entropy(cls.float(), t.tgt_cls, reduction="none")).sum() / t.wsum
    z = torch.zeros_like(loc_loss)
    loss = torch.where(t.none_matched, loc_loss, loc_loss + 10 * box_loss + cls_loss + iou_loss)
    return loss


def fwd_bwd():
    for x in (loc, iou, box, cls):
        x.grad = None
    loss_all().backward()


count(lambda: loss_all(), "loss arithmetic, forward")
count(fwd_bwd, "loss arithmetic, forward+backward")
print(f"rows: candidates R = {R}, positions B*P = {B * P}")
