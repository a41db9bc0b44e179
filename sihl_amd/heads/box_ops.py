"""Complete-IoU matching cost and loss (torchvision.ops.complete_box_iou / complete_box_iou_loss,
published definition - SURVEY.md App. B).  Loss-side arithmetic stays in PyTorch-ROCm device ops
(autograd plumbing); the kernels carry the conv/MLP cost."""
import math

import torch
from torch import Tensor


def sq(x: Tensor) -> Tensor:
    """x * x.  Same value as squaring by ``pow``, another autograd node: the backward of ``pow(x, 2)`` is
    ``2 * grad * x.pow(1)``, and ATen computes ``pow(x, 1)`` as a CLONE - a device-to-device memcpy - which inside a captured
    training step becomes a graph memcpy node.  The four such nodes of the detection loss were what set the faulting graph
    of profiles/r02_graph_fault_experiments.txt apart from the passing one besides its kernels; losses on this path square
    by multiplication, so a captured step holds no copy node of theirs."""
    return x * x


def complete_box_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 1e-7) -> Tensor:
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[None, :, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = inter / (a1[:, None] + a2[None, :] - inter)
    elt = torch.min(boxes1[:, None, :2], boxes2[None, :, :2])
    erb = torch.max(boxes1[:, None, 2:], boxes2[None, :, 2:])
    ewh = (erb - elt).clamp(min=0)
    diag2 = sq(ewh[..., 0]) + sq(ewh[..., 1]) + eps
    c1 = (boxes1[:, :2] + boxes1[:, 2:]) / 2
    c2 = (boxes2[:, :2] + boxes2[:, 2:]) / 2
    dist2 = sq(c1[:, None, :] - c2[None, :, :]).sum(-1)
    w1, h1 = boxes1[:, None, 2] - boxes1[:, None, 0], boxes1[:, None, 3] - boxes1[:, None, 1]
    w2, h2 = boxes2[None, :, 2] - boxes2[None, :, 0], boxes2[None, :, 3] - boxes2[None, :, 1]
    v = (4 / math.pi ** 2) * sq(torch.atan(w1 / h1) - torch.atan(w2 / h2))
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return iou - dist2 / diag2 - alpha * v


def complete_box_iou_loss(b1: Tensor, b2: Tensor, eps: float = 1e-7) -> Tensor:
    x1, y1, x2, y2 = b1.unbind(-1)
    x1g, y1g, x2g, y2g = b2.unbind(-1)
    xk1, yk1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xk2, yk2 = torch.min(x2, x2g), torch.min(y2, y2g)
    valid = (yk2 > yk1) & (xk2 > xk1)
    inter = torch.where(valid, (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    diag2 = sq(torch.max(x2, x2g) - torch.min(x1, x1g)) + sq(torch.max(y2, y2g) - torch.min(y1, y1g)) + eps
    dist2 = sq(((x1 + x2) - (x1g + x2g)) / 2) + sq(((y1 + y2) - (y1g + y2g)) / 2)
    v = (4 / math.pi ** 2) * sq(torch.atan((x2g - x1g) / (y2g - y1g)) - torch.atan((x2 - x1) / (y2 - y1)))
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return 1 - iou + dist2 / diag2 + alpha * v
