"""Oracle (CPU, pure torch) restatement of the reference's dense heads.

TEST INFRASTRUCTURE ONLY.  Reference files followed (relative to
/root/reference/src/sihl):
  heads/object_detection.py:14-217,252-284   ObjectDetection (+ bbox_matching)
  heads/semantic_segmentation.py:11-92       SemanticSegmentation
  heads/semantic_segmentation.py:123-182     SPPM, UAFM
  heads/multiclass_classification.py:47-52   MulticlassClassification (config-1 plumbing)
  heads/instance_segmentation.py:15-278      InstanceSegmentation (CondInst mask decode)
  heads/depth_estimation.py:15-122           DepthEstimation (AdaBins on the SemanticSegmentation decoder; 8f rank 4)
  heads/keypoint_detection.py:15-322,342-378 KeypointDetection (FCPose dynamic heatmaps; 8f rank 4)
  heads/quadrilateral_detection.py:13-211,258-324  QuadrilateralDetection (8f rank 4); torchvision's
                                             sigmoid_focal_loss restated from its published definition (unpinned)
torchvision 0.21 ``ops.complete_box_iou`` / ``complete_box_iou_loss`` are NOT in
the container; they are restated from the published CIoU definition (SURVEY.md
App. B) and are "parity unpinned".
"""
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from oracle.layers import (ConvNormAct, Conv2dNormActivation, Interpolate, MLP,
                           SequentialConvBlocks, SimpleUpscaler)


# --------------------------------------------------------------------------- CIoU
def _pairwise_iou_parts(b1: Tensor, b2: Tensor):
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area1[:, None] + area2[None, :] - inter
    return inter / union, union


def complete_box_iou(boxes1: Tensor, boxes2: Tensor, eps: float = 1e-7) -> Tensor:
    """(N,4) x (M,4) xyxy -> (N,M) CIoU = IoU - rho^2/c^2 - alpha*v."""
    iou, _ = _pairwise_iou_parts(boxes1, boxes2)
    lti = torch.min(boxes1[:, None, :2], boxes2[None, :, :2])
    rbi = torch.max(boxes1[:, None, 2:], boxes2[None, :, 2:])
    whi = (rbi - lti).clamp(min=0)
    diag2 = whi[..., 0] ** 2 + whi[..., 1] ** 2 + eps
    cx1 = (boxes1[:, 0] + boxes1[:, 2]) / 2
    cy1 = (boxes1[:, 1] + boxes1[:, 3]) / 2
    cx2 = (boxes2[:, 0] + boxes2[:, 2]) / 2
    cy2 = (boxes2[:, 1] + boxes2[:, 3]) / 2
    dist2 = (cx1[:, None] - cx2[None, :]) ** 2 + (cy1[:, None] - cy2[None, :]) ** 2
    diou = iou - dist2 / diag2
    w1 = boxes1[:, None, 2] - boxes1[:, None, 0]
    h1 = boxes1[:, None, 3] - boxes1[:, None, 1]
    w2 = boxes2[None, :, 2] - boxes2[None, :, 0]
    h2 = boxes2[None, :, 3] - boxes2[None, :, 1]
    v = (4 / math.pi ** 2) * (torch.atan(w1 / h1) - torch.atan(w2 / h2)) ** 2
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return diou - alpha * v


def complete_box_iou_loss(b1: Tensor, b2: Tensor, eps: float = 1e-7) -> Tensor:
    """Elementwise (N,4),(N,4) -> (N,) loss = 1 - IoU + rho^2/c^2 + alpha*v (reduction none)."""
    x1, y1, x2, y2 = b1.unbind(-1)
    x1g, y1g, x2g, y2g = b2.unbind(-1)
    xk1, yk1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xk2, yk2 = torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.zeros_like(x1)
    m = (yk2 > yk1) & (xk2 > xk1)
    inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    diag2 = (xc2 - xc1) ** 2 + (yc2 - yc1) ** 2 + eps
    dist2 = (((x1 + x2) - (x1g + x2g)) / 2) ** 2 + (((y1 + y2) - (y1g + y2g)) / 2) ** 2
    diou_loss = 1 - iou + dist2 / diag2
    v = (4 / math.pi ** 2) * (torch.atan((x2g - x1g) / (y2g - y1g)) - torch.atan((x2 - x1) / (y2 - y1))) ** 2
    with torch.no_grad():
        alpha = v / (1 - iou + v + eps)
    return diou_loss + alpha * v


# --------------------------------------------------------------------------- OD head
def anchor_offsets_and_scales(sizes: List[Tuple[int, int]], device=None) -> Tuple[Tensor, Tensor]:
    """Cell centres (cx,cy,cx,cy) and half-cell (-1/2w,-1/2h,1/2w,1/2h) per level, concatenated
    (object_detection.py:83-97).  ``sizes`` = [(h, w)] for bottom..top level."""
    offs, scls = [], []
    for h, w in sizes:
        y0, x0 = 1 / h / 2, 1 / w / 2
        ys = torch.linspace(y0, 1 - y0, steps=h, device=device)
        xs = torch.linspace(x0, 1 - x0, steps=w, device=device)
        gx = xs[None, :].expand(h, w).reshape(-1)
        gy = ys[:, None].expand(h, w).reshape(-1)
        offs.append(torch.stack([gx, gy, gx, gy], dim=1))
        scls.append(torch.tensor([-x0, -y0, x0, y0], device=device).expand(h * w, 4))
    return torch.cat(offs), torch.cat(scls)


def _island(t: Tensor) -> Tensor:
    """The reference's fp32 loss islands (`.to(torch.float32)` under a disabled autocast, object_detection.py:158-208).
    A float64 tensor stays float64: tests/test_gpu_fullsize.py runs this oracle in double precision as the yardstick that
    measures the fp32 noise floor; in fp32 / bf16 the cast is the reference's."""
    return t if t.dtype == torch.float64 else t.float()


def bbox_matching(anchors: Tensor, gt_boxes: Tensor, topk: int, relative: bool = False):
    """Top-k-per-GT one-to-many assignment (object_detection.py:252-284)."""
    A, G = anchors.shape[0], gt_boxes.shape[0]
    assign = torch.full((A,), -1, device=anchors.device)
    o2m = torch.zeros((A,), device=anchors.device, dtype=anchors.dtype)
    if G == 0:
        return assign, o2m
    ious = complete_box_iou(anchors, gt_boxes).clamp(0)
    top_v, top_i = torch.topk(ious, k=topk, dim=0)
    in_topk = torch.zeros((A, G), dtype=torch.bool, device=anchors.device)
    in_topk.scatter_(0, top_i, True)
    best_iou, best_gt = torch.max(ious * in_topk.to(ious.dtype), dim=1)
    valid = in_topk.any(dim=1)
    assign[valid] = best_gt[valid]
    if not relative:
        o2m[valid] = best_iou[valid]
        return assign, o2m
    denom = top_v[0][best_gt]
    o2m[valid] = (best_iou[valid] / denom[valid]).nan_to_num(0)
    return assign, o2m


class ObjectDetection(nn.Module):
    """Anchor-free detector: per-level 1x1 conv+BN laterals, four shared MLPs, top-k decode."""

    def __init__(self, in_channels: List[int], num_classes: int, bottom_level: int = 3,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 4,
                 max_instances: int = 100):
        assert num_classes > 0 and len(in_channels) > top_level
        assert 0 < bottom_level <= top_level and num_channels % 4 == 0
        assert num_layers >= 0 and max_instances > 0
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        self.laterals = nn.ModuleList([
            Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=None)
            for l in self.levels])
        hidden = [num_channels] * num_layers

        def mlp(out):
            return MLP(num_channels, hidden + [out], norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)

        self.loc_head = mlp(1)
        self.loc_head[-2].bias.data.fill_(-5.0)  # object_detection.py:58
        self.cls_head = mlp(num_classes)
        self.box_head = mlp(4)
        self.iou_head = mlp(1)
        self.output_shapes = {
            "num_instances": ("batch_size",),
            "scores": ("batch_size", max_instances),
            "classes": ("batch_size", max_instances),
            "boxes": ("batch_size", max_instances, 4),
        }

    bbox_matching = staticmethod(bbox_matching)

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        feats = [lat(inputs[l]) for l, lat in zip(self.levels, self.laterals)]
        return torch.cat([f.flatten(2).transpose(1, 2) for f in feats], dim=1)  # (B, P, C)

    def get_offsets_and_scales(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor]:
        return anchor_offsets_and_scales([tuple(inputs[l].shape[2:]) for l in self.levels],
                                         device=inputs[0].device)

    def forward(self, inputs: List[Tensor]):
        B, _, H, W = inputs[0].shape
        full = torch.tensor([[[W, H, W, H]]], device=inputs[0].device)
        flat = self._flat_feats(inputs)
        offsets, scales = self.get_offsets_and_scales(inputs)
        loc_logits, idx = self.loc_head(flat).squeeze(2).topk(self.max_instances, dim=1)
        rows = torch.arange(B)[:, None].expand(B, self.max_instances)
        sel = flat[rows, idx]
        scores = loc_logits.sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        classes = self.cls_head(sel).max(dim=2).indices
        boxes = (offsets[idx] + scales[idx] * self.box_head(sel).exp()) * full
        return num_instances, scores, classes, boxes

    def training_step(self, inputs: List[Tensor], classes: List[Tensor], boxes: List[Tensor],
                      is_validating: bool = False):
        assert len(inputs) > self.top_level
        B, _, H, W = inputs[0].shape
        full = torch.tensor([[W, H, W, H]], device=inputs[0].device)
        offsets, scales = self.get_offsets_and_scales(inputs)
        anchors = (offsets + scales) * full
        matches = [bbox_matching(anchors, boxes[b], self.topk, relative=True) for b in range(B)]
        assignment = torch.stack([m[0] for m in matches])
        rel_iou = torch.stack([m[1] for m in matches])
        flat = self._flat_feats(inputs)

        loc_logits = self.loc_head(flat).squeeze(2)
        loc_target = (rel_iou == 1.0).to(_island(loc_logits).dtype)
        loc_loss = F.binary_cross_entropy_with_logits(_island(loc_logits), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        if rel_iou.max() == 0:  # object_detection.py:165-172
            z = torch.zeros_like(loc_loss)
            return loc_loss, {"location_loss": loc_loss, "box_loss": z, "class_loss": z, "iou_loss": z}

        iou_preds = self.iou_head(flat).squeeze(2)
        iou_loss = F.mse_loss(_island(iou_preds), rel_iou, reduction="none").sum() / rel_iou.sum()

        mask = rel_iou > 0
        wts = rel_iou[mask]
        sel = flat[mask]
        off_sel = torch.cat([offsets[m] for m in mask])
        scl_sel = torch.cat([scales[m] for m in mask])
        box_preds = off_sel + scl_sel * self.box_head(sel).exp()
        box_target = torch.cat([boxes[b][assignment[b, m]] for b, m in enumerate(mask)])
        box_loss = complete_box_iou_loss(_island(box_preds), _island(box_target) / full)
        box_loss = (wts * box_loss).sum() / wts.sum()

        cls_logits = self.cls_head(sel)
        cls_target = torch.cat([classes[b][assignment[b, m]] for b, m in enumerate(mask)])
        cls_loss = F.cross_entropy(_island(cls_logits), cls_target, reduction="none")
        cls_loss = (wts * cls_loss).sum() / wts.sum()

        loss = loc_loss + 10 * box_loss + cls_loss + iou_loss
        return loss, {"location_loss": loc_loss, "box_loss": box_loss,
                      "class_loss": cls_loss, "iou_loss": iou_loss}


# --------------------------------------------------------------------------- instance segmentation (CondInst)
def masks_to_boxes(masks: Tensor) -> Tensor:
    """torchvision.ops.masks_to_boxes: (N, H, W) masks -> (N, 4) xyxy boxes of their non-zero pixels."""
    out = torch.zeros((masks.shape[0], 4), device=masks.device, dtype=torch.float)
    for i, m in enumerate(masks):
        ys, xs = torch.where(m != 0)
        out[i] = torch.stack([xs.min(), ys.min(), xs.max(), ys.max()]).float()
    return out


def grid_offsets(sizes: List[Tuple[int, int]], device=None) -> List[Tensor]:
    """Per map the (h, w, 2) grid of normalised cell centres (x, y) (instance_segmentation.py:87-104)."""
    out = []
    for h, w in sizes:
        ys = torch.linspace(1 / h / 2, 1 - 1 / h / 2, steps=h, device=device)
        xs = torch.linspace(1 / w / 2, 1 - 1 / w / 2, steps=w, device=device)
        out.append(torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=2))
    return out


def dynamic_mask_net(feats: Tensor, weights: Tensor, c: int) -> Tensor:
    """The per-instance 3-layer 1x1 network of CondInst: feats (n, c+2, h, w), weights (n, (c+2)c + c + cc + c + c + 1)
    -> sigmoid masks (n, h, w) (instance_segmentation.py:139-157 / 248-260)."""
    n = weights.shape[0]
    i = 0
    w1 = weights[:, i: (i := i + (c + 2) * c)].reshape(n, c + 2, c)
    b1 = weights[:, i: (i := i + c)].reshape(n, c, 1, 1)
    w2 = weights[:, i: (i := i + c * c)].reshape(n, c, c)
    b2 = weights[:, i: (i := i + c)].reshape(n, c, 1, 1)
    w3 = weights[:, i: (i := i + c)].reshape(n, c, 1)
    b3 = weights[:, i:].reshape(n, 1, 1, 1)
    x = F.silu(torch.einsum("nchw,ncd->ndhw", feats, w1) + b1)
    x = F.silu(torch.einsum("nchw,ncd->ndhw", x, w2) + b2)
    x = torch.einsum("nchw,ncd->ndhw", x, w3) + b3
    return x.squeeze(1).sigmoid()


class InstanceSegmentation(nn.Module):
    """Anchor-free instance segmentation: the detector's laterals / location / class MLPs plus a kernel MLP that
    emits, per instance, the 169 parameters of a tiny 1x1 network run over 8 mask-feature channels + 2 relative
    coordinates of the mask level."""

    def __init__(self, in_channels: List[int], num_classes: int, mask_level: int = 3, bottom_level: int = 3,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 4, max_instances: int = 100):
        assert num_classes > 0 and len(in_channels) > top_level
        assert 0 < bottom_level <= top_level and num_channels % 4 == 0
        assert num_layers >= 0 and max_instances > 0
        super().__init__()
        self.in_channels, self.num_classes, self.mask_level = in_channels, num_classes, mask_level
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        self.laterals = nn.ModuleList([
            Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=None) for l in self.levels])
        hidden = [num_channels] * num_layers

        def mlp(out):
            return MLP(num_channels, hidden + [out], norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)

        self.loc_head = mlp(1)
        self.loc_head[-2].bias.data.fill_(-5.0)
        self.cls_head = mlp(num_classes)
        c = self.mask_num_channels = 8
        self.kernel_head = mlp((c + 2) * c + c + c * c + c + c + 1)
        self.mask_lateral = Conv2dNormActivation(in_channels[mask_level], num_channels, 1, activation_layer=None)
        self.mask_head = Conv2dNormActivation(num_channels, c, 3, activation_layer=nn.SiLU)
        scale = 2 ** bottom_level
        self.output_shapes = {"num_instances": ("batch_size",), "scores": ("batch_size", max_instances),
                              "classes": ("batch_size", max_instances),
                              "masks": ("batch_size", max_instances, f"height/{scale}", f"width/{scale}")}

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        feats = [lat(inputs[l]) for l, lat in zip(self.levels, self.laterals)]
        return torch.cat([f.flatten(2).transpose(1, 2) for f in feats], dim=1)

    def _sizes(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def forward(self, inputs: List[Tensor]):
        B, _, H, W = inputs[0].shape
        K, c, dev = self.max_instances, self.mask_num_channels, inputs[0].device
        flat = self._flat_feats(inputs)
        loc_logits, idx = self.loc_head(flat).squeeze(2).topk(K, dim=1)
        rows = torch.arange(B)[:, None].expand(B, K)
        scores = loc_logits.sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        sel = flat[rows, idx]
        mask_feats = self.mask_head(self.mask_lateral(inputs[self.mask_level]))  # (B, c, h, w)
        h, w = mask_feats.shape[2:]
        offsets = torch.cat([g.reshape(-1, 2) for g in grid_offsets(self._sizes(inputs), dev)])[idx]  # (B, K, 2)
        grid = grid_offsets([(h, w)], dev)[0].permute(2, 0, 1)  # (2, h, w)
        rel = grid[None, None] - offsets[:, :, :, None, None]
        feats = torch.cat([mask_feats[:, None].expand(B, K, c, h, w), rel], dim=2)
        masks = dynamic_mask_net(feats.reshape(B * K, c + 2, h, w), self.kernel_head(sel).reshape(B * K, -1), c)
        classes = self.cls_head(sel).max(dim=2).indices
        masks = F.interpolate(masks.reshape(B, K, h, w), size=(H, W), mode="bilinear")
        return num_instances, scores, classes, masks

    def training_step(self, inputs: List[Tensor], classes: List[Tensor], masks: List[Tensor],
                      is_validating: bool = False):
        assert len(inputs) > self.top_level
        dev = inputs[0].device
        B, _, H, W = inputs[0].shape
        c = self.mask_num_channels
        valid = [m.any((1, 2)) if m.shape[0] > 0 else None for m in masks]  # drop empty masks (:178-181)
        classes = [cl[v] for v, cl in zip(valid, classes)]
        masks = [m[v] for v, m in zip(valid, masks)]
        grids = grid_offsets(self._sizes(inputs), dev)
        centres = torch.cat([g.reshape(-1, 2) for g in grids])  # (P, 2)
        half = torch.cat([torch.tensor([-0.5 / w_, -0.5 / h_, 0.5 / w_, 0.5 / h_], device=dev).expand(h_ * w_, 4)
                          for h_, w_ in self._sizes(inputs)])
        anchors = (centres.repeat(1, 2) + half) * torch.tensor([[W, H, W, H]], device=dev)
        boxes = [masks_to_boxes(m) for m in masks]
        matches = [bbox_matching(anchors, boxes[b], self.topk, relative=True) for b in range(B)]
        assignment = torch.stack([m[0] for m in matches])
        rel_iou = torch.stack([m[1] for m in matches])

        flat = self._flat_feats(inputs)
        o2m = rel_iou > 0
        wts = rel_iou[o2m]
        sel = flat[o2m]
        loc_logits = self.loc_head(flat).squeeze(2)
        loc_target = (rel_iou == 1.0).to(torch.float32)
        loc_loss = F.binary_cross_entropy_with_logits(loc_logits.float(), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        z = torch.zeros_like(loc_loss)
        if rel_iou.max() == 0:
            return loc_loss, {"location_loss": loc_loss, "mask_loss": z, "class_loss": z}

        mask_feats = self.mask_head(self.mask_lateral(inputs[self.mask_level]))
        h, w = mask_feats.shape[2:]
        grid = grid_offsets([(h, w)], dev)[0].permute(2, 0, 1)
        per_image = []
        for b in range(B):
            idx = o2m[b].nonzero()[:, 0]
            if idx.numel():
                rel = grid[None] - centres[idx][:, :, None, None]
                per_image.append(torch.cat([mask_feats[b][None].expand(idx.numel(), c, h, w), rel], dim=1))
        preds = dynamic_mask_net(torch.cat(per_image), self.kernel_head(sel), c)
        target = torch.cat([masks[b][assignment[b, m]] for b, m in enumerate(o2m) if m.any()]).to(preds)
        target = F.interpolate(target.unsqueeze(1), size=preds.shape[1:], mode="bilinear").squeeze(1)
        num = (preds * target).sum((1, 2))
        den = (preds ** 2 + target ** 2).sum((1, 2))
        mask_loss = 1 - 2 * num.float() / den
        mask_loss = (wts * mask_loss).sum() / wts.sum()
        cls_target = torch.cat([classes[b][assignment[b, m]] for b, m in enumerate(o2m) if m.any()])
        cls_loss = F.cross_entropy(self.cls_head(sel).float(), cls_target, reduction="none")
        cls_loss = (wts * cls_loss).sum() / wts.sum()
        loss = loc_loss + 10 * mask_loss + cls_loss
        return loss, {"location_loss": loc_loss, "mask_loss": mask_loss, "class_loss": cls_loss}


# --------------------------------------------------------------------------- keypoint detection (FCPose)
def dynamic_keypoint_net(feats: Tensor, weights: Tensor, c: int, k: int) -> Tensor:
    """feats (n, c+2, h, w), weights (n, (c+2)c + c + cc + c + ck + k) -> heatmap logits (n, k, h, w)
    (keypoint_detection.py:139-158 / 276-288)."""
    n = weights.shape[0]
    i = 0
    w1 = weights[:, i: (i := i + (c + 2) * c)].reshape(n, c + 2, c)
    b1 = weights[:, i: (i := i + c)].reshape(n, c, 1, 1)
    w2 = weights[:, i: (i := i + c * c)].reshape(n, c, c)
    b2 = weights[:, i: (i := i + c)].reshape(n, c, 1, 1)
    w3 = weights[:, i: (i := i + c * k)].reshape(n, c, k)
    b3 = weights[:, i:].reshape(n, k, 1, 1)
    x = F.silu(torch.einsum("nchw,ncd->ndhw", feats, w1) + b1)
    x = F.silu(torch.einsum("nchw,ncd->ndhw", x, w2) + b2)
    return torch.einsum("nchw,ncd->ndhw", x, w3) + b3


def keypoints_to_boxes(keypoints: Tensor, presence: Tensor) -> Tensor:
    """(n, k, 2) keypoints, (n, k) bool presence -> (n, 4) xyxy hull of the present keypoints (:342-353)."""
    lo = keypoints.masked_fill(~presence[..., None], float("inf")).amin(dim=1)
    hi = keypoints.masked_fill(~presence[..., None], float("-inf")).amax(dim=1)
    return torch.cat([lo, hi], dim=-1)


def keypoints_to_heatmaps(keypoints: Tensor, presence: Tensor, height: int, width: int, img_height: int,
                          img_width: int) -> Tensor:
    """One-hot (n, k, height, width) targets at the rounded, rescaled keypoint cell; all-zero where absent (:355-378)."""
    xs = (keypoints[:, :, 0] * ((width - 1) / (img_width - 1))).clamp(0, width - 1).round().to(torch.int64)
    ys = (keypoints[:, :, 1] * ((height - 1) / (img_height - 1))).clamp(0, height - 1).round().to(torch.int64)
    gx = F.one_hot(xs, width).to(torch.float32)
    gy = F.one_hot(ys, height).to(torch.float32)
    return presence[:, :, None, None] * gx.unsqueeze(2) * gy.unsqueeze(3)


class KeypointDetection(nn.Module):
    """Instance-aware keypoint heatmaps: the detector front end + a kernel MLP emitting a per-instance 3-layer 1x1
    network (c+2 -> c -> c -> num_keypoints, c = 32) run over the mask-level features + relative coordinates."""

    def __init__(self, in_channels: List[int], num_keypoints: int, mask_level: int = 3, bottom_level: int = 5,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 4, max_instances: int = 100):
        assert num_keypoints > 0 and len(in_channels) > top_level
        assert 0 < bottom_level <= top_level and num_channels % 4 == 0
        assert num_layers >= 0 and max_instances > 0
        super().__init__()
        self.in_channels, self.num_keypoints, self.mask_level = in_channels, num_keypoints, mask_level
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        self.laterals = nn.ModuleList([
            Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=None) for l in self.levels])
        hidden = [num_channels] * num_layers

        def mlp(out):
            return MLP(num_channels, hidden + [out], norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)

        self.loc_head = mlp(1)
        self.loc_head[-2].bias.data.fill_(-5.0)
        self.presence_head = mlp(num_keypoints)
        c = self.mask_num_channels = 32
        self.kernel_head = mlp((c + 2) * c + c + c * c + c + c * num_keypoints + num_keypoints)
        self.mask_lateral = Conv2dNormActivation(in_channels[mask_level], num_channels, 1, activation_layer=None)
        self.mask_head = Conv2dNormActivation(num_channels, c, 3, activation_layer=nn.SiLU)
        self.output_shapes = {"num_instances": ("batch_size",), "scores": ("batch_size", max_instances),
                              "presence": ("batch_size", max_instances, num_keypoints),
                              "keypoints": ("batch_size", max_instances, num_keypoints, 2)}

    keypoints_to_boxes = staticmethod(keypoints_to_boxes)
    keypoints_to_heatmaps = staticmethod(keypoints_to_heatmaps)

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        feats = [lat(inputs[l]) for l, lat in zip(self.levels, self.laterals)]
        return torch.cat([f.flatten(2).transpose(1, 2) for f in feats], dim=1)

    def _sizes(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def forward(self, inputs: List[Tensor], output_heatmaps: bool = False):
        B, _, H, W = inputs[0].shape
        K, c, dev = self.max_instances, self.mask_num_channels, inputs[0].device
        flat = self._flat_feats(inputs)
        loc_logits, idx = self.loc_head(flat).squeeze(2).topk(K, dim=1)
        rows = torch.arange(B)[:, None].expand(B, K)
        scores = loc_logits.sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        sel = flat[rows, idx]
        mask_feats = self.mask_head(self.mask_lateral(inputs[self.mask_level]))
        h, w = mask_feats.shape[2:]
        offsets = torch.cat([g.reshape(-1, 2) for g in grid_offsets(self._sizes(inputs), dev)])[idx]
        grid = grid_offsets([(h, w)], dev)[0].permute(2, 0, 1)
        rel = grid[None, None] - offsets[:, :, :, None, None]
        feats = torch.cat([mask_feats[:, None].expand(B, K, c, h, w), rel], dim=2).reshape(B * K, c + 2, h, w)
        heat = dynamic_keypoint_net(feats, self.kernel_head(sel).reshape(B * K, -1), c, self.num_keypoints)
        heat = heat.reshape(B, K, self.num_keypoints, h, w)
        presence = self.presence_head(sel).sigmoid()
        if output_heatmaps:
            return heat.flatten(3, 4).softmax(3).reshape(heat.shape)
        flat_idx = heat.flatten(3, 4).max(3).indices
        ky, kx = flat_idx // h, flat_idx % h  # the reference divides by the mask HEIGHT for both (:165)
        ky = (ky.float() + 0.5) / h * H
        kx = (kx.float() + 0.5) / w * W
        return num_instances, scores, presence, torch.stack([kx, ky], dim=3)

    def training_step(self, inputs: List[Tensor], presence: List[Tensor], keypoints: List[Tensor],
                      is_validating: bool = False):
        assert len(inputs) > self.top_level
        dev = inputs[0].device
        B, _, H, W = inputs[0].shape
        c, nk = self.mask_num_channels, self.num_keypoints
        keep = [p.any(dim=1) for p in presence]  # instances without any visible keypoint are dropped (:186-192)
        keypoints = [k[m] for k, m in zip(keypoints, keep)]
        presence = [p[m] for p, m in zip(presence, keep)]
        boxes = [keypoints_to_boxes(k, p) for k, p in zip(keypoints, presence)]
        centres = torch.cat([g.reshape(-1, 2) for g in grid_offsets(self._sizes(inputs), dev)])
        half = torch.cat([torch.tensor([-0.5 / w_, -0.5 / h_, 0.5 / w_, 0.5 / h_], device=dev).expand(h_ * w_, 4)
                          for h_, w_ in self._sizes(inputs)])
        anchors = (centres.repeat(1, 2) + half) * torch.tensor([[W, H, W, H]], device=dev)
        matches = [bbox_matching(anchors, boxes[b], self.topk, relative=True) for b in range(B)]
        assignment = torch.stack([m[0] for m in matches])
        rel_iou = torch.stack([m[1] for m in matches])
        flat = self._flat_feats(inputs)
        o2m = rel_iou > 0
        wts = rel_iou[o2m].reshape(-1, 1)
        sel = flat[o2m]
        loc_logits = self.loc_head(flat).squeeze(2)
        loc_target = (rel_iou == 1.0).to(torch.float32)
        loc_loss = F.binary_cross_entropy_with_logits(loc_logits.float(), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        z = torch.zeros_like(loc_loss)
        if rel_iou.max() == 0:
            return loc_loss, {"location_loss": loc_loss, "keypoint_loss": z, "presence_loss": z}

        target_presence = torch.cat([presence[b][assignment[b, o2m[b]]] for b in range(B)])
        presence_loss = F.binary_cross_entropy_with_logits(self.presence_head(sel).float(),
                                                           target_presence.to(torch.float32), reduction="none")
        presence_loss = (wts * presence_loss).sum() / wts.sum()

        mask_feats = self.mask_head(self.mask_lateral(inputs[self.mask_level]))
        h, w = mask_feats.shape[2:]
        grid = grid_offsets([(h, w)], dev)[0].permute(2, 0, 1)
        per_image = []
        for b in range(B):
            idx = o2m[b].nonzero()[:, 0]
            if idx.numel():
                rel = grid[None] - centres[idx][:, :, None, None]
                per_image.append(torch.cat([mask_feats[b][None].expand(idx.numel(), c, h, w), rel], dim=1))
        heat = dynamic_keypoint_net(torch.cat(per_image), self.kernel_head(sel), c, nk)
        target_kpts = torch.cat([keypoints[b][assignment[b, o2m[b]]] for b in range(B)])
        target_heat = keypoints_to_heatmaps(target_kpts, target_presence, h, w, H, W)
        kp_loss = F.cross_entropy(heat.flatten(2).transpose(1, 2).float(), target_heat.flatten(2).transpose(1, 2),
                                  reduction="none")  # classes = the h*w cells, "spatial" dim = keypoints
        kp_loss = (wts * kp_loss).sum() / wts.sum()
        loss = loc_loss + kp_loss + presence_loss
        return loss, {"location_loss": loc_loss, "keypoint_loss": kp_loss, "presence_loss": presence_loss}


# --------------------------------------------------------------------------- quadrilateral detection
def sigmoid_focal_loss(logits: Tensor, targets: Tensor, alpha: float = 0.25, gamma: float = 2.0) -> Tensor:
    """torchvision.ops.sigmoid_focal_loss(reduction="none"): RetinaNet focal loss on logits."""
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    return (alpha * targets + (1 - alpha) * (1 - targets)) * loss


def quad_bbox_matching(anchors: Tensor, gt_boxes: Tensor, topk: int):
    """Top-k-per-GT assignment without the clamp of the box detector, plus the one-to-one (best anchor per GT) mask
    (quadrilateral_detection.py:258-289)."""
    A, G = anchors.shape[0], gt_boxes.shape[0]
    dev = anchors.device
    assign = torch.full((A,), -1, device=dev)
    o2o = torch.zeros((A,), dtype=torch.bool, device=dev)
    iou_out, rel = torch.zeros((A,), device=dev), torch.zeros((A,), device=dev)
    if G == 0:
        return assign, o2o, iou_out, rel
    ious = complete_box_iou(anchors, gt_boxes)
    top_v, top_i = torch.topk(ious, k=topk, dim=0)
    best = torch.zeros((A, G), dtype=torch.bool, device=dev).scatter_(0, top_i[0:1], True)
    in_topk = torch.zeros((A, G), dtype=torch.bool, device=dev).scatter_(0, top_i, True)
    max_iou, max_gt = torch.max(ious * in_topk.float(), dim=1)
    valid = in_topk.any(dim=1)
    assign = torch.where(valid, max_gt, assign)
    iou_out = torch.where(valid, max_iou, iou_out)
    rel = torch.where(valid, (max_iou / top_v[0][max_gt]).nan_to_num(0), rel)
    return assign, best.any(dim=1), iou_out, rel


def canonicalize_and_convexify(quads: Tensor) -> Tensor:
    """Vertices sorted by angle around the centroid; concave vertices replaced by the midpoint of their neighbours
    (quadrilateral_detection.py:291-313)."""
    rel = quads - quads.mean(dim=1, keepdim=True)
    order = torch.atan2(rel[..., 1], rel[..., 0]).sort(dim=1).indices
    v = torch.gather(quads, 1, order[..., None].expand(-1, -1, 2))
    nxt, prv = v[:, [1, 2, 3, 0]], v[:, [3, 0, 1, 2]]
    cross = (nxt[..., 0] - v[..., 0]) * (prv[..., 1] - v[..., 1]) - (nxt[..., 1] - v[..., 1]) * (prv[..., 0] - v[..., 0])
    return torch.where((cross < 0)[..., None], (prv + nxt) * 0.5, v)


def quads_to_boxes(quads: Tensor) -> Tensor:
    x, y = quads[..., 0], quads[..., 1]
    return torch.stack([x.min(-1).values, y.min(-1).values, x.max(-1).values, y.max(-1).values], 1)


class QuadrilateralDetection(nn.Module):
    """Detector skeleton with conv+BN+SiLU laterals plus a global-context vector (1x1 conv+BN+SiLU of the top level,
    globally averaged) added to every position; quads = cell centre + tanh(MLP) per vertex."""

    def __init__(self, in_channels: List[int], num_classes: int, bottom_level: int = 3, top_level: int = 5,
                 num_channels: int = 256, num_layers: int = 4, max_instances: int = 100):
        assert num_classes > 0 and len(in_channels) > top_level
        assert 0 < bottom_level <= top_level and num_channels % 4 == 0
        assert num_layers >= 0 and max_instances > 0
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        self.laterals = nn.ModuleList([
            Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=nn.SiLU) for l in self.levels])
        self.global_context = nn.Sequential(
            Conv2dNormActivation(in_channels[top_level], num_channels, 1, activation_layer=nn.SiLU),
            nn.AdaptiveAvgPool2d(1))
        hidden = [num_channels] * num_layers

        def mlp(out):
            return MLP(num_channels, hidden + [out], norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)

        self.loc_head, self.class_head, self.quad_head = mlp(1), mlp(num_classes), mlp(8)
        self.output_shapes = {"num_instances": ("batch_size",), "scores": ("batch_size", max_instances),
                              "classes": ("batch_size", max_instances), "quads": ("batch_size", max_instances, 4, 2)}

    bbox_matching = staticmethod(quad_bbox_matching)
    canonicalize_and_convexify = staticmethod(canonicalize_and_convexify)
    quads_to_boxes = staticmethod(quads_to_boxes)

    def _sizes(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def get_offsets_and_levels(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor]:
        dev = inputs[0].device
        centres = torch.cat([g.reshape(-1, 2) for g in grid_offsets(self._sizes(inputs), dev)])
        levels = torch.cat([torch.full((h * w, 1), l, device=dev) for l, (h, w) in zip(self.levels, self._sizes(inputs))])
        return centres.repeat(1, 4), levels

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        ctx = self.global_context(inputs[self.top_level])
        feats = [lat(inputs[l]) + ctx for l, lat in zip(self.levels, self.laterals)]
        return torch.cat([f.flatten(2).transpose(1, 2) for f in feats], dim=1)

    def forward(self, inputs: List[Tensor]):
        B, _, H, W = inputs[0].shape
        K = self.max_instances
        flat = self._flat_feats(inputs)
        offsets, _ = self.get_offsets_and_levels(inputs)
        loc_logits, idx = self.loc_head(flat).squeeze(2).topk(K, dim=1)
        rows = torch.arange(B)[:, None].expand(B, K)
        scores = loc_logits.sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        sel = flat[rows, idx]
        quads = (offsets[idx] + self.quad_head(sel).tanh()) * torch.tensor([[[W, H] * 4]], device=flat.device)
        classes = self.class_head(sel).max(dim=2).indices
        return num_instances, scores, classes, quads.reshape(B, K, 4, 2)

    def training_step(self, inputs: List[Tensor], classes: List[Tensor], quads: List[Tensor],
                      is_validating: bool = False):
        assert len(inputs) > self.top_level
        dev = inputs[0].device
        B, _, H, W = inputs[0].shape
        flat = self._flat_feats(inputs)
        offsets, levels = self.get_offsets_and_levels(inputs)
        scale = torch.sigmoid(levels - self.top_level)
        anchors = (offsets[:, :4] + torch.tensor([[-1, -1, 1, 1]], device=dev) * scale) * torch.tensor([[W, H] * 2], device=dev)
        matches = [quad_bbox_matching(anchors, quads_to_boxes(q), self.topk) for q in quads]
        assignment = torch.stack([m[0] for m in matches])
        o2o = torch.stack([m[1] for m in matches])
        rel_iou = torch.stack([m[3] for m in matches])
        o2m = rel_iou > 0
        loc_target = rel_iou / self.topk
        loc_target[o2o] = 1
        wts = rel_iou[o2m]
        sel = flat[o2m]
        off_sel = torch.cat([offsets[m] for m in o2m])
        quad_preds = (off_sel + self.quad_head(sel).tanh()).clamp(0, 1).reshape(-1, 4, 2)
        quad_target = torch.cat([quads[b][assignment[b, m]] for b, m in enumerate(o2m)])
        quad_target = canonicalize_and_convexify(quad_target) / torch.tensor([[[W, H]] * 4], device=dev)
        quad_loss = F.l1_loss(quad_preds.float(), quad_target, reduction="none").sum(dim=(1, 2))
        quad_loss = 10 * (wts * quad_loss).sum() / wts.sum()
        cls_target = torch.cat([classes[b][assignment[b, m]] for b, m in enumerate(o2m)])
        cls_loss = sigmoid_focal_loss(self.class_head(sel).float(),
                                      F.one_hot(cls_target, self.num_classes).to(torch.float32)).sum(dim=1)
        cls_loss = 10 * (wts * cls_loss).sum() / wts.sum()
        loc_logits = self.loc_head(flat).squeeze(2)
        loc_loss = F.binary_cross_entropy_with_logits(loc_logits.float(), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        loss = loc_loss + quad_loss + cls_loss
        return loss, {"location_loss": loc_loss, "quad_loss": quad_loss, "class_loss": cls_loss}


# --------------------------------------------------------------------------- SemSeg head
class SPPM(nn.Module):
    """Bilinear "pooling" pyramid: resize to p x p, 1x1 ConvNormAct, resize back, sum, 1x1
    (semantic_segmentation.py:123-160)."""

    def __init__(self, in_channels, out_channels, pool_sizes=(1, 2, 4), with_shortcut=False):
        super().__init__()
        self.with_shortcut = with_shortcut
        self.pools = nn.ModuleList(
            [nn.Sequential(Interpolate(size=p), ConvNormAct(in_channels, out_channels, 1))
             for p in pool_sizes] if len(pool_sizes) > 0 else [nn.Identity()])
        if with_shortcut:
            self.shortcut = ConvNormAct(in_channels, out_channels, 1)
        self.out_conv = ConvNormAct(out_channels, out_channels, 1)

    def forward(self, x: Tensor) -> Tensor:
        size = x.shape[2:]
        acc = None
        for pool in self.pools:
            y = F.interpolate(pool(x), size=size, mode="bilinear")
            acc = y if acc is None else acc + y
        if self.with_shortcut:
            acc = acc + self.shortcut(x)
        return self.out_conv(acc)


class UAFM(nn.Module):
    """alpha = sigmoid(conv3x3([mean_c x1, max_c x1, mean_c x2, max_c x2])); x1*alpha + x2*(1-alpha)
    (semantic_segmentation.py:163-182)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = ConvNormAct(4, 1, norm=None, act="sigmoid")

    def forward(self, x1: Tensor, x2: Tensor) -> Tensor:
        stats = torch.cat([x1.mean(1, keepdim=True), x1.amax(1, keepdim=True),
                           x2.mean(1, keepdim=True), x2.amax(1, keepdim=True)], dim=1)
        a = self.conv(stats)
        return x1 * a + x2 * (1 - a)


class SemanticSegmentation(nn.Module):
    """PP-LiteSeg style decoder (semantic_segmentation.py:11-92)."""

    def __init__(self, in_channels: List[int], num_classes: int, bottom_level: int = 3,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 3,
                 pool_sizes=(1, 2, 4), ignore_index=None):
        assert num_classes > 0 and len(in_channels) > top_level >= bottom_level > 0
        assert num_channels > 0 and num_layers >= 0
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        self.num_channels, self.num_layers = num_channels, num_layers
        self.pool_sizes = tuple(pool_sizes)
        self.bottom_level, self.top_level = bottom_level, top_level
        self.ignore_index = ignore_index or -100  # semantic_segmentation.py:51 (0 is un-ignorable)
        self.levels = list(range(bottom_level, top_level + 1))
        self.rev_levels = list(reversed(range(bottom_level, top_level)))
        self.context_aggregation = SPPM(in_channels[top_level], num_channels, self.pool_sizes)
        self.lateral_convs = nn.ModuleList([ConvNormAct(in_channels[l], num_channels) for l in self.rev_levels])
        self.upscalers = nn.ModuleList([SimpleUpscaler(num_channels, num_channels) for _ in self.rev_levels])
        self.fusions = nn.ModuleList([UAFM(num_channels, num_channels) for _ in self.rev_levels])
        self.out_conv = nn.Sequential(SequentialConvBlocks(num_channels, num_channels, num_layers),
                                      nn.Conv2d(num_channels, num_classes, kernel_size=1))
        self.output_shapes = {"score_maps": ("batch_size", "height", "width"),
                              "class_maps": ("batch_size", "height", "width")}

    def get_logits(self, inputs: List[Tensor]) -> Tensor:
        x = self.context_aggregation(inputs[self.top_level])
        for l, lat, up, fuse in zip(self.rev_levels, self.lateral_convs, self.upscalers, self.fusions):
            x = fuse(lat(inputs[l]), up(x))
        return self.out_conv(x)

    def forward(self, inputs: List[Tensor]):
        x = F.interpolate(self.get_logits(inputs), size=inputs[0].shape[2:])  # nearest
        return x.softmax(dim=1).max(dim=1)

    def training_step(self, inputs: List[Tensor], targets: Tensor):
        logits = F.interpolate(self.get_logits(inputs), size=targets.shape[1:])
        return F.cross_entropy(logits, targets, ignore_index=self.ignore_index), {}


# --------------------------------------------------------------------------- depth estimation (AdaBins)
DEPTH_EPS = 1e-5  # sihl.utils.EPS (utils/__init__.py:20)


class DepthEstimation(SemanticSegmentation):
    """Per-image adaptive bins (a conv tower + global average on the top level gives bin widths) weighted by the
    per-pixel bin scores of the segmentation decoder (depth_estimation.py:15-122)."""

    def __init__(self, in_channels: List[int], lower_bound: float, upper_bound: float, bottom_level: int = 3,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 1, num_bins: int = 256):
        assert lower_bound < upper_bound and num_bins > 1 and num_layers > 0
        super().__init__(in_channels=in_channels, num_classes=num_bins, num_channels=num_channels,
                         bottom_level=bottom_level, top_level=top_level, num_layers=num_layers)
        self.num_bins, self.lower_bound, self.upper_bound = num_bins, lower_bound, upper_bound
        self.bin_head = nn.Sequential(SequentialConvBlocks(in_channels[top_level], num_channels, num_layers),
                                      nn.Conv2d(num_channels, num_bins, kernel_size=1), nn.AdaptiveAvgPool2d(1),
                                      nn.Flatten())
        stride = 2 ** bottom_level
        self.output_shapes = {"depth_maps": ("batch_size", f"height/{stride}", f"width/{stride}")}

    def normalize(self, x: Tensor) -> Tensor:
        return (x - self.lower_bound) / (self.upper_bound - self.lower_bound)

    def denormalize(self, x: Tensor) -> Tensor:
        return x * (self.upper_bound - self.lower_bound) + self.lower_bound

    def get_bin_centers(self, inputs: List[Tensor]) -> Tensor:
        widths = self.bin_head(inputs[self.top_level]).relu() + DEPTH_EPS
        widths = widths / widths.sum(dim=1, keepdim=True)
        return widths.cumsum(dim=1) - widths / 2

    def get_depth_map(self, inputs: List[Tensor], bin_centers: Tensor) -> Tensor:
        weights = self.get_logits(inputs).relu() + DEPTH_EPS
        weights = weights / weights.sum(dim=1, keepdim=True)
        return (bin_centers[:, :, None, None] * weights).sum(dim=1, keepdim=True).clamp(0, 1)

    def forward(self, inputs: List[Tensor]) -> Tensor:
        depth = self.denormalize(self.get_depth_map(inputs, self.get_bin_centers(inputs)))
        return F.interpolate(depth, size=inputs[0].shape[2:]).squeeze(1)

    def training_step(self, inputs: List[Tensor], targets: Tensor, masks: Tensor):
        B = targets.shape[0]
        masks, targets = masks[:, None], self.normalize(targets[:, None])
        centers = self.get_bin_centers(inputs)
        depth = self.get_depth_map(inputs, centers)
        pred_shape = depth.shape[2:]
        depth = F.interpolate(depth, size=targets.shape[2:])
        g = (depth[masks] + DEPTH_EPS).log() - (targets[masks] + DEPTH_EPS).log()
        pix_loss = torch.sqrt(g.var() + 0.15 * g.mean().pow(2)) * 10  # scale-invariant log loss
        masks = F.interpolate(masks.to(torch.uint8), size=pred_shape, mode="nearest") > 0
        targets = F.interpolate(targets, size=pred_shape)
        hist = []
        for b in range(B):  # bidirectional chamfer distance between bin centres and the valid target depths
            dist = (centers[b][None, :] - targets[b][masks[b]][:, None]).pow(2)
            hist.append(dist.min(dim=1).values.mean() + dist.min(dim=0).values.mean())
        hist_loss = torch.stack(hist).mean()
        return pix_loss + hist_loss, {"pixel_loss": pix_loss, "hist_loss": hist_loss}


# --------------------------------------------------------------------------- config-1 plumbing head
class MulticlassClassification(nn.Module):
    """conv blocks on one level -> 1x1 -> global average -> logits
    (multiclass_classification.py:11-69); stock PyTorch only (BASELINE config 1)."""

    def __init__(self, in_channels: List[int], num_classes: int, num_channels: int = 256,
                 num_layers: int = 1, level: int = 5, label_smoothing: float = 0.0):
        assert num_classes > 0 and len(in_channels) > level and num_channels > 0 and num_layers > 0
        super().__init__()
        self.level, self.num_classes, self.label_smoothing = level, num_classes, label_smoothing
        self.convs = nn.Sequential(
            SequentialConvBlocks(in_channels[level], num_channels, num_layers),
            nn.Conv2d(num_channels, num_classes, kernel_size=1),
            nn.AdaptiveAvgPool2d(1), nn.Flatten())
        self.output_shapes = {"scores": ("batch_size", num_classes), "classes": ("batch_size",)}

    def forward(self, inputs: List[Tensor]):
        return self.convs(inputs[self.level]).softmax(dim=1).max(dim=1)

    def training_step(self, inputs: List[Tensor], target: Tensor):
        logits = self.convs(inputs[self.level])
        return F.cross_entropy(logits, target.to(logits.device), label_smoothing=self.label_smoothing), {}
