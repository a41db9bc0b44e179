"""QuadrilateralDetection head on the HIP path (reference src/sihl/heads/quadrilateral_detection.py:13-324;
SURVEY 8f rank 4).

The detector skeleton of ``ObjectDetection`` with two differences the reference makes: the laterals are
conv+BN+SiLU and a global-context vector (1x1 conv+BN+SiLU of the top level, averaged over the map, :60-62,
:113-118) is added to every position.  On the HIP kernels: the laterals and the context conv (fused conv + BatchNorm
+ SiLU blocks), the location MLP over all positions, per-image top-K, row gather, class / vertex MLPs on the K rows.
The vertex arithmetic (cell centre + tanh), the matching and the three losses are fp32 device ops.
"""
from functools import partial
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from sihl_amd import ops
from sihl_amd.heads.box_ops import complete_box_iou
from sihl_amd.heads.mlp import MLP, forward_many
from sihl_amd.layers.convblocks import Conv2dNormActivation


def sigmoid_focal_loss(logits: Tensor, targets: Tensor, alpha: float = 0.25, gamma: float = 2.0) -> Tensor:
    """RetinaNet focal loss on logits, unreduced (what the reference gets from torchvision.ops, :225-227)."""
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    q = 1 - p_t
    mod = q * q if gamma == 2.0 else q ** gamma  # (x * x, not pow(x, 2): no clone node in a captured backward - box_ops.sq)
    return (alpha * targets + (1 - alpha) * (1 - targets)) * ce * mod


class QuadrilateralDetection(nn.Module):
    def __init__(self, in_channels: List[int], num_classes: int, bottom_level: int = 3, top_level: int = 5,
                 num_channels: int = 256, num_layers: int = 4, max_instances: int = 100) -> None:
        assert num_classes > 0, num_classes
        assert len(in_channels) > top_level, (len(in_channels), top_level)
        assert 0 < bottom_level <= top_level, (bottom_level, top_level)
        assert num_channels % 4 == 0, num_channels
        assert num_layers >= 0, num_layers
        assert max_instances > 0, max_instances
        super().__init__()
        self.in_channels, self.num_classes = in_channels, num_classes
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        mlp = partial(MLP, norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)
        conv = partial(Conv2dNormActivation, activation_layer=nn.SiLU)
        self.laterals = nn.ModuleList([conv(in_channels[l], num_channels, 1) for l in self.levels])
        self.global_context = nn.Sequential(conv(in_channels[top_level], num_channels, 1), nn.AdaptiveAvgPool2d(1))
        hidden = [num_channels] * num_layers
        self.loc_head = mlp(num_channels, hidden + [1])
        self.class_head = mlp(num_channels, hidden + [num_classes])
        self.quad_head = mlp(num_channels, hidden + [8])
        self.output_shapes = {"num_instances": ("batch_size",), "scores": ("batch_size", max_instances),
                              "classes": ("batch_size", max_instances), "quads": ("batch_size", max_instances, 4, 2)}

    # ------------------------------------------------------------------ helpers
    def _level_hw(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def get_offsets_and_levels(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor]:
        """(P, 8) cell centres repeated for the four vertices and (P, 1) level numbers (:92-111)."""
        device = inputs[self.bottom_level].device
        level_hw = self._level_hw(inputs)
        offsets, _ = ops.od_anchors(level_hw, device)  # (P, 4) = (cx, cy, cx, cy)
        levels = torch.cat([torch.full((h * w, 1), l, device=device) for l, (h, w) in zip(self.levels, level_hw)])
        return offsets.repeat(1, 2), levels

    def _context_nhwc(self, inputs: List[Tensor]) -> Tensor:
        """(B, 1, 1, C): conv+BN+SiLU of the top level, averaged over the map."""
        return self.global_context[0].forward_nhwc(ops.nhwc(inputs[self.top_level])).float().mean(dim=(1, 2), keepdim=True)

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        """(B, P, C) lateral features + context, positions in level-major, row-major order (:113-125)."""
        ctx = self._context_nhwc(inputs)
        feats = [lat.forward_nhwc(ops.nhwc(inputs[l])) for l, lat in zip(self.levels, self.laterals)]
        B, C = feats[0].shape[0], feats[0].shape[-1]
        return torch.cat([(f + ctx.to(f.dtype)).reshape(B, -1, C) for f in feats], dim=1)

    def get_features(self, inputs: List[Tensor]) -> List[Tensor]:
        ctx = self._context_nhwc(inputs)
        return [ops.nchw_view(lat.forward_nhwc(ops.nhwc(inputs[l])) + ctx.to(inputs[l].dtype))
                for l, lat in zip(self.levels, self.laterals)]

    def get_saliency(self, inputs: List[Tensor]) -> Tensor:
        B, _, fh, fw = inputs[self.bottom_level].shape
        out = torch.zeros((B, fh, fw), device=inputs[self.bottom_level].device)
        ctx = self._context_nhwc(inputs)
        for lat, l in zip(self.laterals, self.levels):
            h, w = inputs[l].shape[2:]
            f = lat.forward_nhwc(ops.nhwc(inputs[l]))
            f = (f + ctx.to(f.dtype)).reshape(B * h * w, -1)
            s = self.loc_head(f).float().sigmoid().reshape(B, 1, h, w)
            out = torch.maximum(out, F.interpolate(s, size=(fh, fw)).squeeze(1))
        return out

    # ------------------------------------------------------------------ inference
    def forward(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        B, _, H, W = inputs[0].shape
        flat = self._flat_feats(inputs)
        device = flat.device
        P, K = flat.shape[1], self.max_instances
        loc_logits = self.loc_head(flat.view(B * P, -1))
        top_vals, top_idx = ops.topk_rows(loc_logits, B, P, K, estride=loc_logits.stride(0))
        sel = ops.gather_rows(flat, top_idx).view(B * K, -1)
        scores = top_vals.float().sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        offsets, _ = self.get_offsets_and_levels(inputs)
        quad_raw, cls_logits = forward_many([self.quad_head, self.class_head], sel)  # one launch in inference
        quads = offsets[top_idx.long().reshape(-1)] + quad_raw.float().tanh()
        quads = quads * torch.tensor([[W, H] * 4], device=device, dtype=torch.float32)
        classes = cls_logits.float().reshape(B, K, -1).max(dim=2).indices
        return num_instances, scores, classes, quads.reshape(B, K, 4, 2)

    # ------------------------------------------------------------------ training
    def training_step(self, inputs: List[Tensor], classes: List[Tensor], quads: List[Tensor],
                      is_validating: bool = False) -> Tuple[Tensor, Dict[str, float]]:
        assert len(inputs) > self.top_level, "too few input levels"
        B, _, H, W = inputs[0].shape
        flat = self._flat_feats(inputs)
        device = flat.device
        P, C = flat.shape[1], flat.shape[2]
        classes = [c.to(device) for c in classes]
        quads = [q.to(device=device, dtype=torch.float32) for q in quads]

        offsets, levels = self.get_offsets_and_levels(inputs)
        directions = torch.tensor([[-1.0, -1.0, 1.0, 1.0]], device=device)
        scale = torch.sigmoid((levels - self.top_level).float())
        anchors = (offsets[:, :4] + directions * scale) * torch.tensor([[W, H] * 2], device=device, dtype=torch.float32)
        matches = [self.bbox_matching(anchors, self.quads_to_boxes(q), self.topk) for q in quads]
        assignment = torch.stack([m[0] for m in matches])
        o2o = torch.stack([m[1] for m in matches])
        rel_iou = torch.stack([m[3] for m in matches])

        o2m = rel_iou > 0
        loc_target = rel_iou / self.topk
        loc_target[o2o] = 1
        wts = rel_iou[o2m]
        pos = o2m.nonzero()  # (n, 2): image, position - row-major, the order of the reference's per-image concatenation
        sel = flat[o2m]
        gt_of = assignment[o2m]

        quad_preds = (offsets[pos[:, 1]] + self.quad_head(sel).float().tanh()).clamp(0, 1).reshape(-1, 4, 2)
        counts = torch.tensor([0] + [q.shape[0] for q in quads[:-1]], device=device).cumsum(0)
        flat_gt = counts[pos[:, 0]] + gt_of
        quad_target = self.canonicalize_and_convexify(torch.cat(quads)[flat_gt])
        quad_target = quad_target / torch.tensor([[[W, H]] * 4], device=device, dtype=torch.float32)
        quad_loss = F.l1_loss(quad_preds, quad_target, reduction="none").sum(dim=(1, 2))
        quad_loss = 10 * (wts * quad_loss).sum() / wts.sum()

        cls_target = F.one_hot(torch.cat(classes)[flat_gt], self.num_classes).to(torch.float32)
        cls_loss = sigmoid_focal_loss(self.class_head(sel).float(), cls_target).sum(dim=1)
        cls_loss = 10 * (wts * cls_loss).sum() / wts.sum()

        loc_logits = self.loc_head(flat.view(B * P, C)).reshape(B, P)
        loc_loss = F.binary_cross_entropy_with_logits(loc_logits.float(), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        loss = loc_loss + quad_loss + cls_loss
        return loss, {"location_loss": loc_loss, "quad_loss": quad_loss, "class_loss": cls_loss}

    @staticmethod
    def bbox_matching(anchors: Tensor, gt_boxes: Tensor, topk: int):
        """Top-k-per-GT assignment (no IoU clamp, unlike the box detector) and the one-to-one mask = best anchor of
        each ground truth (:258-289).  Returns (assignment, o2o mask, IoU, relative IoU), each (A,)."""
        A, G = anchors.shape[0], gt_boxes.shape[0]
        device = anchors.device
        assign = torch.full((A,), -1, device=device)
        zeros = torch.zeros((A,), device=device)
        if G == 0:
            return assign, torch.zeros((A,), dtype=torch.bool, device=device), zeros, zeros
        ious = complete_box_iou(anchors, gt_boxes.to(anchors.dtype))
        top_v, top_i = torch.topk(ious, k=topk, dim=0)
        o2o = torch.zeros((A, G), dtype=torch.bool, device=device).scatter_(0, top_i[0:1], True).any(dim=1)
        in_topk = torch.zeros((A, G), dtype=torch.bool, device=device).scatter_(0, top_i, True)
        best_iou, best_gt = torch.max(ious * in_topk.float(), dim=1)
        valid = in_topk.any(dim=1)
        rel = (best_iou / top_v[0][best_gt]).nan_to_num(0)
        return (torch.where(valid, best_gt, assign), o2o, torch.where(valid, best_iou, zeros),
                torch.where(valid, rel, zeros))

    @staticmethod
    def canonicalize_and_convexify(quads: Tensor) -> Tensor:
        """Vertices ordered by angle around the centroid; a concave vertex moves to the midpoint of its neighbours
        (:291-313)."""
        rel = quads - quads.mean(dim=1, keepdim=True)
        order = torch.atan2(rel[..., 1], rel[..., 0]).sort(dim=1).indices
        v = torch.gather(quads, 1, order[..., None].expand(-1, -1, 2))
        nxt, prv = v[:, [1, 2, 3, 0]], v[:, [3, 0, 1, 2]]
        cross = ((nxt[..., 0] - v[..., 0]) * (prv[..., 1] - v[..., 1])
                 - (nxt[..., 1] - v[..., 1]) * (prv[..., 0] - v[..., 0]))
        return torch.where((cross < 0)[..., None], (prv + nxt) * 0.5, v)

    @staticmethod
    def quads_to_boxes(quads: Tensor) -> Tensor:
        x, y = quads[..., 0], quads[..., 1]
        return torch.stack([x.min(-1).values, y.min(-1).values, x.max(-1).values, y.max(-1).values], 1)

    # ------------------------------------------------------------------ validation
    def on_validation_start(self) -> None:
        from sihl_amd.metrics import BoxMeanAveragePrecision

        self._val_losses: List[Tensor] = []
        self.map_computer = BoxMeanAveragePrecision([1, min(self.max_instances, 10), self.max_instances])

    def validation_step(self, inputs, classes, quads):
        """Reference :236-256: box mAP on the axis-aligned hulls of predicted and target quadrilaterals."""
        _, scores, pred_classes, pred_quads = self.forward(inputs)
        B, K = pred_quads.shape[:2]
        pred_boxes = self.quads_to_boxes(pred_quads.reshape(B * K, 4, 2)).reshape(B, K, 4)
        self.map_computer.update(
            [{"scores": s, "labels": c, "boxes": b} for s, c, b in zip(scores, pred_classes, pred_boxes)],
            [{"labels": c, "boxes": self.quads_to_boxes(q)} for c, q in zip(classes, quads)])
        loss, metrics = self.training_step(inputs, classes, quads, is_validating=True)
        self._val_losses.append(loss.detach())
        return loss, metrics

    def on_validation_end(self) -> Dict[str, float]:
        metrics = self.map_computer.compute() if hasattr(self, "map_computer") else {}
        metrics["loss"] = torch.stack(self._val_losses).mean().item() if self._val_losses else float("nan")
        return metrics
