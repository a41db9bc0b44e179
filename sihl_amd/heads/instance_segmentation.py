"""InstanceSegmentation head on the HIP path (reference src/sihl/heads/instance_segmentation.py:15-278;
SURVEY 8f rank 1: the mask decode).

forward: the detector's front end (1x1 conv+BN laterals into the flat (B, P, C) buffer, location MLP over all
positions, per-image top-K, gather) + class and kernel MLPs on the K rows + mask branch (1x1 conv+BN lateral,
3x3 conv+BN+SiLU to 8 channels, both matrix-core conv blocks) + ONE kernel for the CondInst decode: per-instance
10->8->8->1 network over the mask features and relative coordinates, sigmoid, bilinear resize to the input size.
The (B, K, 10, h, w) feature stack, the three einsum results and the low-resolution masks of the reference are never
written to HBM.

training_step: same modules (HIP forward and backward through autograd); the one-to-many matching, the per-object
dynamic network on the gathered rows and the dice / cross-entropy losses are fp32 device ops as in the reference's
autocast-disabled islands.
"""
from functools import partial
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from sihl_amd import ops
from sihl_amd.heads.mlp import MLP, forward_many
from sihl_amd.heads.object_detection import ObjectDetection
from sihl_amd.layers.convblocks import Conv2dNormActivation


def masks_to_boxes(masks: Tensor) -> Tensor:
    """(N, H, W) non-empty masks -> (N, 4) xyxy boxes of their non-zero pixels (torchvision.ops.masks_to_boxes)."""
    n, H, W = masks.shape
    nz = masks != 0
    cols, rows = nz.any(dim=1), nz.any(dim=2)  # (N, W), (N, H)
    xs, ys = torch.arange(W, device=masks.device), torch.arange(H, device=masks.device)
    big = max(H, W)
    x0 = torch.where(cols, xs, big).min(dim=1).values
    x1 = torch.where(cols, xs, -1).max(dim=1).values
    y0 = torch.where(rows, ys, big).min(dim=1).values
    y1 = torch.where(rows, ys, -1).max(dim=1).values
    return torch.stack([x0, y0, x1, y1], dim=1).float()


def dynamic_mask_net(feats: Tensor, weights: Tensor, c: int) -> Tensor:
    """feats (n, c+2, h, w), weights (n, 169) -> sigmoid masks (n, h, w): the differentiable (training) form of the
    network the decode kernel evaluates (reference :248-260)."""
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
    return (torch.einsum("nchw,ncd->ndhw", x, w3) + b3).squeeze(1).sigmoid()


class InstanceSegmentation(nn.Module):
    def __init__(self, in_channels: List[int], num_classes: int, mask_level: int = 3, bottom_level: int = 3,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 4, max_instances: int = 100) -> None:
        assert num_classes > 0, num_classes
        assert len(in_channels) > top_level, (len(in_channels), top_level)
        assert 0 < bottom_level <= top_level, (bottom_level, top_level)
        assert num_channels % 4 == 0, num_channels
        assert num_layers >= 0, num_layers
        assert max_instances > 0, max_instances
        super().__init__()
        self.in_channels, self.num_classes, self.mask_level = in_channels, num_classes, mask_level
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        mlp = partial(MLP, norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)
        self.laterals = nn.ModuleList(
            [Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=None) for l in self.levels])
        hidden = [num_channels] * num_layers
        self.loc_head = mlp(num_channels, hidden + [1])
        self.loc_head[-2].bias.data.fill_(-5.0)  # reference :66
        self.cls_head = mlp(num_channels, hidden + [num_classes])
        c = self.mask_num_channels = 8
        self.kernel_head = mlp(num_channels, hidden + [(c + 2) * c + c + c * c + c + c + 1])
        self.mask_lateral = Conv2dNormActivation(in_channels[mask_level], num_channels, 1, activation_layer=None)
        self.mask_head = Conv2dNormActivation(num_channels, c, 3, activation_layer=nn.SiLU)
        scale = 2 ** bottom_level
        self.output_shapes = {"num_instances": ("batch_size",), "scores": ("batch_size", max_instances),
                              "classes": ("batch_size", max_instances),
                              "masks": ("batch_size", max_instances, f"height/{scale}", f"width/{scale}")}

    # ------------------------------------------------------------------ helpers
    def _level_hw(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        feats = [lat.forward_nhwc(ops.nhwc(inputs[l])) for l, lat in zip(self.levels, self.laterals)]
        B, C = feats[0].shape[0], feats[0].shape[-1]
        return torch.cat([f.reshape(B, -1, C) for f in feats], dim=1)

    def _mask_feats_nhwc(self, inputs: List[Tensor]) -> Tensor:
        return self.mask_head.forward_nhwc(self.mask_lateral.forward_nhwc(ops.nhwc(inputs[self.mask_level])))

    def get_saliency(self, inputs: List[Tensor]) -> Tensor:
        return self.forward(inputs)[3].amax(dim=1)

    # ------------------------------------------------------------------ inference
    def forward(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        B, _, H, W = inputs[0].shape
        flat = self._flat_feats(inputs)
        P, K = flat.shape[1], self.max_instances
        loc_logits = self.loc_head(flat.view(B * P, -1))
        top_vals, top_idx = ops.topk_rows(loc_logits, B, P, K, estride=loc_logits.stride(0))
        sel = ops.gather_rows(flat, top_idx).view(B * K, -1)
        scores = top_vals.sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        cls_logits, kernels = forward_many([self.cls_head, self.kernel_head], sel)  # one launch in inference
        classes = cls_logits.reshape(B, K, -1).float().argmax(dim=2)
        masks = ops.iseg_mask_decode(self._mask_feats_nhwc(inputs), kernels, top_idx,
                                     self._level_hw(inputs), (H, W))
        return num_instances, scores, classes, masks

    # ------------------------------------------------------------------ training
    def training_step(self, inputs: List[Tensor], classes: List[Tensor], masks: List[Tensor],
                      is_validating: bool = False) -> Tuple[Tensor, Dict[str, float]]:
        assert len(inputs) > self.top_level, "too few input levels"
        device = inputs[self.bottom_level].device
        B, _, H, W = inputs[0].shape
        c = self.mask_num_channels
        masks = [m.to(device) for m in masks]
        classes = [k.to(device) for k in classes]
        valid = [m.any((1, 2)) if m.shape[0] > 0 else None for m in masks]  # empty masks are dropped (:178-181)
        classes = [k[v] for v, k in zip(valid, classes)]
        masks = [m[v] for v, m in zip(valid, masks)]

        offsets, scales = ops.od_anchors(self._level_hw(inputs), device)  # (P, 4): (cx, cy, cx, cy), half cells
        anchors = (offsets + scales) * torch.tensor([[W, H, W, H]], device=device, dtype=torch.float32)
        boxes = [masks_to_boxes(m) for m in masks]
        assignment, rel_iou = ObjectDetection.batched_matching(anchors, boxes, self.topk)

        flat = self._flat_feats(inputs)
        P, C = flat.shape[1], flat.shape[2]
        loc_logits = self.loc_head(flat.view(B * P, C)).reshape(B, P)
        loc_target = (rel_iou == 1.0).to(torch.float32)
        loc_loss = F.binary_cross_entropy_with_logits(loc_logits.float(), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        z = torch.zeros_like(loc_loss)
        if rel_iou.max() == 0:
            return loc_loss, {"location_loss": loc_loss, "mask_loss": z, "class_loss": z}

        o2m = rel_iou > 0
        wts = rel_iou[o2m]
        pos = o2m.nonzero()  # (n, 2): image, position - row-major like the boolean indexing of the reference
        sel = flat[o2m]
        mask_feats = ops.nchw_view(self._mask_feats_nhwc(inputs)).float()  # (B, c, h, w)
        h, w = mask_feats.shape[2:]
        ys = (torch.arange(h, device=device, dtype=torch.float32) + 0.5) / h
        xs = (torch.arange(w, device=device, dtype=torch.float32) + 0.5) / w
        grid = torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)])  # (2, h, w)
        rel = grid[None] - offsets[pos[:, 1], :2][:, :, None, None]
        feats = torch.cat([mask_feats[pos[:, 0]], rel], dim=1)  # (n, c + 2, h, w)
        preds = dynamic_mask_net(feats, self.kernel_head(sel).float(), c)

        gt_idx = assignment[o2m]
        counts = torch.tensor([0] + [m.shape[0] for m in masks[:-1]], device=device).cumsum(0)
        flat_gt = counts[pos[:, 0]] + gt_idx
        target = torch.cat(masks).to(preds)[flat_gt]
        target = F.interpolate(target.unsqueeze(1), size=preds.shape[1:], mode="bilinear").squeeze(1)
        num = (preds * target).sum((1, 2))
        den = (preds * preds + target * target).sum((1, 2))
        mask_loss = 1 - 2 * num.float() / den
        mask_loss = (wts * mask_loss).sum() / wts.sum()

        cls_loss = F.cross_entropy(self.cls_head(sel).float(), torch.cat(classes)[flat_gt], reduction="none")
        cls_loss = (wts * cls_loss).sum() / wts.sum()
        loss = loc_loss + 10 * mask_loss + cls_loss
        return loss, {"location_loss": loc_loss, "mask_loss": mask_loss, "class_loss": cls_loss}

    def on_validation_start(self) -> None:
        from sihl_amd.metrics import MaskMeanAveragePrecision

        self._val_losses: List[Tensor] = []
        self.map_computer = MaskMeanAveragePrecision([1, min(self.max_instances, 10), self.max_instances])

    def validation_step(self, inputs, classes, masks):
        """Reference :308-320: mask mAP (COCO protocol, iou_type "segm") of ``forward``'s detections, thresholded at 0.5,
        plus the training loss on the same inputs.  The per-image IoU matrices are computed on the device."""
        with torch.no_grad():
            _, scores, pred_classes, pred_masks = self.forward(inputs)
        self.map_computer.update(
            [{"scores": s, "labels": c, "masks": m > 0.5} for s, c, m in zip(scores, pred_classes, pred_masks)],
            [{"labels": c, "masks": m > 0.5} for c, m in zip(classes, masks)])
        loss, metrics = self.training_step(inputs, classes, masks, is_validating=True)
        self._val_losses.append(loss.detach())
        return loss, metrics

    def on_validation_end(self) -> Dict[str, float]:
        metrics = self.map_computer.compute() if hasattr(self, "map_computer") else {}
        metrics["loss"] = torch.nanmean(torch.stack(self._val_losses).float()).item() if self._val_losses else float("nan")
        return metrics
