"""ObjectDetection head on the HIP path (reference src/sihl/heads/object_detection.py:14-217,252-284).

forward: per-level 1x1 conv+BN laterals written straight into the flat (B, P, C) position buffer ->
loc MLP over all P positions -> per-image top-K (LDS bitonic sort) -> gather K rows -> class / box
MLPs on K rows -> fused sigmoid / argmax / closed-form-anchor box decode.
training_step: same laterals + MLPs (matrix-core kernels with hand-written backward); anchor
matching and the four losses are fp32 device ops, as in the reference's autocast-disabled islands.
"""
import os
from functools import partial
from types import SimpleNamespace
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from sihl_amd import ops
from sihl_amd.heads.box_ops import complete_box_iou, complete_box_iou_loss
from sihl_amd.heads.mlp import MLP, forward_many
from sihl_amd.layers.convblocks import Conv2dNormActivation


FUSED_LOSS = True  # test / A-B switch: False = the loss as PyTorch device ops (the formulas of the reference, line by line)
CAT_LATERALS = bool(os.environ.get("SIHL_CAT_LATERALS"))  # test / A-B switch (read once): True = per-level outputs + torch.cat


class ObjectDetection(nn.Module):
    def __init__(self, in_channels: List[int], num_classes: int, bottom_level: int = 3, top_level: int = 5,
                 num_channels: int = 256, num_layers: int = 4, max_instances: int = 100) -> None:
        assert num_classes > 0, num_classes
        assert len(in_channels) > top_level, (len(in_channels), top_level)
        assert 0 < bottom_level <= top_level, (bottom_level, top_level)
        assert num_channels % 4 == 0, num_channels
        assert num_layers >= 0, num_layers
        assert max_instances > 0, max_instances
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances = max_instances
        self.topk = 9
        mlp = partial(MLP, norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)
        self.laterals = nn.ModuleList(
            [Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=None) for l in self.levels])
        hidden = [num_channels] * num_layers
        self.loc_head = mlp(num_channels, hidden + [1])
        self.loc_head[-2].bias.data.fill_(-5.0)  # reference :58
        self.cls_head = mlp(num_channels, hidden + [num_classes])
        self.box_head = mlp(num_channels, hidden + [4])
        self.iou_head = mlp(num_channels, hidden + [1])
        self.output_shapes = {
            "num_instances": ("batch_size",),
            "scores": ("batch_size", max_instances),
            "classes": ("batch_size", max_instances),
            "boxes": ("batch_size", max_instances, 4),
        }

    # ------------------------------------------------------------------ helpers
    def _level_hw(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def get_offsets_and_scales(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor]:
        return ops.od_anchors(self._level_hw(inputs), inputs[self.bottom_level].device)

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        """(B, P, C) lateral features, positions in level-major, row-major order (reference :102-105)."""
        xs = [ops.nhwc(inputs[l]) for l in self.levels]
        if not torch.is_grad_enabled() and not self.training and xs[0].is_cuda and not CAT_LATERALS:
            # inference: every lateral writes its level's rows straight into the flat buffer (image stride P * C)
            B, C = xs[0].shape[0], self.num_channels
            sizes = [x.shape[1] * x.shape[2] for x in xs]
            P = sum(sizes)
            flat = torch.empty((B, P, C), dtype=xs[0].dtype, device=xs[0].device)
            off, ok = 0, True
            for x, lat, n in zip(xs, self.laterals, sizes):
                ok = ok and lat.forward_nhwc_into(x, flat[:, off:off + n], P * C)
                off += n
            if ok:
                return flat
        feats = [lat.forward_nhwc(x) for x, lat in zip(xs, self.laterals)]
        B, C = feats[0].shape[0], feats[0].shape[-1]
        return torch.cat([f.reshape(B, -1, C) for f in feats], dim=1)

    # ------------------------------------------------------------------ inference
    def forward(self, inputs: List[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        B, _, H, W = inputs[0].shape
        level_hw = self._level_hw(inputs)
        flat = self._flat_feats(inputs)
        P, K = flat.shape[1], self.max_instances
        loc_logits = self.loc_head(flat.view(B * P, -1))  # (B*P, 1) view of a vector-padded buffer
        top_vals, top_idx = ops.topk_rows(loc_logits, B, P, K, estride=loc_logits.stride(0))
        sel = ops.gather_rows(flat, top_idx)
        cls_logits, box_raw = forward_many([self.cls_head, self.box_head], sel.view(B * K, -1))  # one launch in inference
        cls_logits, box_raw = cls_logits.reshape(B, K, -1), box_raw.reshape(B, K, 4)
        return ops.od_decode(top_vals, top_idx, cls_logits, box_raw, level_hw, (W, H))

    def get_saliency(self, inputs: List[Tensor]) -> Tensor:
        B, _, fh, fw = inputs[self.bottom_level].shape
        out = torch.zeros((B, fh, fw), device=inputs[self.bottom_level].device)
        for lat, l in zip(self.laterals, self.levels):
            h, w = inputs[l].shape[2:]
            f = lat.forward_nhwc(ops.nhwc(inputs[l]))
            s = self.loc_head(f.reshape(B * h * w, -1)).float().sigmoid().reshape(B, 1, h, w)
            out = torch.maximum(out, F.interpolate(s, size=(fh, fw)).squeeze(1))
        return out

    # ------------------------------------------------------------------ training
    def _targets(self, level_hw, W: int, H: int, classes: List[Tensor], boxes: List[Tensor], device):
        """Everything of the loss that depends on the targets and the anchor grid only - not on the network's output:
        padded ground truths, the matching (reference :139-148, :252-284), the location / IoU targets and the
        fixed-size candidate rows with their weights (see training_step).  The Trainer computes it on the side stream
        while the backbone runs (``prematch``)."""
        t = SimpleNamespace()
        t.full = self._full_size(W, H, device)
        t.offsets, t.scales = ops.od_anchors(level_hw, device)
        anchors = (t.offsets + t.scales) * t.full
        gt, gt_cls, col_ok = self._pad_targets(boxes, classes, device)
        assignment, t.rel_iou, top_i = self._match_padded(anchors, gt, col_ok, self.topk)
        t.num_gt = gt.shape[1]
        t.loc_target = (t.rel_iou == 1.0).to(torch.float32)
        t.loc_norm = t.loc_target.sum()
        if t.num_gt == 0:
            return t
        B, P = t.rel_iou.shape
        G, K = gt.shape[1], top_i.shape[1]
        cand = top_i.permute(0, 2, 1).reshape(B, G * K)  # anchor index of candidate (image, gt, rank)
        cand_gt = torch.arange(G, device=device).repeat_interleave(K)[None, :].expand(B, -1)
        mine = (torch.gather(assignment, 1, cand) == cand_gt) & col_ok.repeat_interleave(K, dim=1)
        t.wts = (torch.gather(t.rel_iou, 1, cand) * mine).reshape(-1)  # 0 for rows the reference does not select
        t.rows = (cand + torch.arange(B, device=device)[:, None] * P).reshape(-1)
        t.anchor_of = cand.reshape(-1)
        t.tgt_box = torch.gather(gt, 1, cand_gt[..., None].expand(-1, -1, 4)).reshape(-1, 4) / t.full
        t.tgt_cls = torch.gather(gt_cls, 1, cand_gt).reshape(-1)
        t.wsum = t.wts.sum()
        t.iou_norm = t.rel_iou.sum()
        t.cand_offsets, t.cand_scales = t.offsets[t.anchor_of], t.scales[t.anchor_of]
        t.none_matched = t.rel_iou.max() == 0  # degenerate ground truths only; stays on the device
        return t

    def prematch(self, image_hw: Tuple[int, int], classes: List[Tensor], boxes: List[Tensor], device) -> None:
        """Compute the target-only part of the next training_step now (the Trainer calls this on its side stream before
        the backbone runs; the level sizes follow the backbone's contract, level l = image / 2^l, and are checked
        against the real feature maps when the result is used)."""
        H, W = int(image_hw[0]), int(image_hw[1])
        level_hw = [(H >> l, W >> l) for l in self.levels]
        self._prematched = (level_hw, (H, W), id(classes), id(boxes), self._targets(level_hw, W, H, classes, boxes, device))

    def training_step(self, inputs: List[Tensor], classes: List[Tensor], boxes: List[Tensor],
                      is_validating: bool = False) -> Tuple[Tensor, Dict[str, float]]:
        """Loss of the reference (:124-217), arranged so that the host never waits for the device: no boolean-mask
        indexing / nonzero / `.max() == 0` branch / pageable host->device copy.  The one-to-many rows (reference
        `rel_iou > 0`, :183-208) are enumerated as the fixed-size candidate list (image, ground truth, rank <
        topk) of the matching itself, each weighted by rel_iou when the anchor's assigned ground truth is that
        one and by exactly 0 otherwise - the same weighted sums over the same rows, in a different order."""
        assert len(inputs) > self.top_level, "too few input levels"
        device = inputs[self.bottom_level].device
        B, _, H, W = inputs[0].shape
        level_hw = self._level_hw(inputs)
        pm, self._prematched = getattr(self, "_prematched", None), None
        if pm is not None and pm[0] == level_hw and pm[1] == (H, W) and pm[2] == id(classes) and pm[3] == id(boxes):
            t = pm[4]
        else:
            t = self._targets(level_hw, W, H, classes, boxes, device)

        flat = self._flat_feats(inputs)
        P, C = flat.shape[1], flat.shape[2]
        loc_logits = self.loc_head(flat.view(B * P, C)).reshape(B, P)

        def location_loss():
            l = F.binary_cross_entropy_with_logits(loc_logits.float(), t.loc_target, reduction="none")
            return l.sum() / t.loc_norm

        if t.num_gt == 0:  # no ground truth in the whole batch (host-side shapes): reference early-out :165-172
            loc_loss = location_loss()
            z = torch.zeros_like(loc_loss)
            return loc_loss, {"location_loss": loc_loss, "box_loss": z, "class_loss": z, "iou_loss": z}

        iou_preds = self.iou_head(flat.view(B * P, C)).reshape(B, P)
        if FUSED_LOSS and flat.is_cuda and t.rel_iou.dtype == torch.float32:
            # the four loss sums and their gradients in one launch (ops.od_loss / sihl_od_loss) instead of ~240 ATen ones
            sel = torch.index_select(flat.view(B * P, C), 0, t.rows)
            out = ops.od_loss(loc_logits, iou_preds, self.box_head(sel), self.cls_head(sel), t)
            return out[0], {"location_loss": out[1], "box_loss": out[2], "class_loss": out[3], "iou_loss": out[4]}
        loc_loss = location_loss()
        z = torch.zeros_like(loc_loss)
        iou_loss = F.mse_loss(iou_preds.float(), t.rel_iou, reduction="none").sum() / t.iou_norm

        sel = torch.index_select(flat.view(B * P, C), 0, t.rows)
        box_preds = t.cand_offsets + t.cand_scales * self.box_head(sel).float().exp()
        box_loss = (t.wts * complete_box_iou_loss(box_preds, t.tgt_box)).sum() / t.wsum

        cls_loss = F.cross_entropy(self.cls_head(sel).float(), t.tgt_cls, reduction="none")
        cls_loss = (t.wts * cls_loss).sum() / t.wsum

        pick = lambda v: torch.where(t.none_matched, z, v)  # noqa: E731
        loss = torch.where(t.none_matched, loc_loss, loc_loss + 10 * box_loss + cls_loss + iou_loss)
        return loss, {"location_loss": loc_loss, "box_loss": pick(box_loss), "class_loss": pick(cls_loss),
                      "iou_loss": pick(iou_loss)}

    def _full_size(self, W: int, H: int, device) -> Tensor:
        """(1, 4) tensor (W, H, W, H), uploaded once per image size (an upload from pageable memory blocks the host
        until the stream has drained)."""
        key = (W, H, str(device))
        cache = self.__dict__.setdefault("_full_cache", {})
        if key not in cache:
            cache[key] = torch.tensor([[W, H, W, H]], device=device, dtype=torch.float32)
        return cache[key]

    @staticmethod
    def _pad_targets(boxes: List[Tensor], classes: List[Tensor], device) -> Tuple[Tensor, Tensor, Tensor]:
        """Per-image target lists -> (B, G, 4) boxes, (B, G) classes, (B, G) validity, G = batch maximum.
        Padded slots get a degenerate-free placeholder box that keeps the CIoU arithmetic finite."""
        B = len(boxes)
        G = max([int(b.shape[0]) for b in boxes], default=0)  # host-side shapes: no device sync
        if G == 0:
            return (torch.zeros((B, 0, 4), device=device), torch.zeros((B, 0), device=device, dtype=torch.int64),
                    torch.zeros((B, 0), device=device, dtype=torch.bool))
        # ONE concatenation per tensor instead of pad_sequence's device copy PER IMAGE (64 tiny copies per step at batch 32):
        # image i contributes its k_i rows followed by G - k_i rows of a constant (placeholder box / class 0 / False); the
        # row counts are host-side shapes, so nothing is uploaded and nothing syncs
        ph_box, ph_cls, ph_true, ph_false = ObjectDetection._padding_constants(device, G)
        parts_b, parts_c, parts_ok = [], [], []
        for b, c in zip(boxes, classes):
            k = int(b.shape[0])
            parts_b += [b.to(device=device, dtype=torch.float32).reshape(-1, 4), ph_box[:G - k]]
            parts_c += [c.to(device=device, dtype=torch.int64).reshape(-1), ph_cls[:G - k]]
            parts_ok += [ph_true[:k], ph_false[:G - k]]
        return torch.cat(parts_b).view(B, G, 4), torch.cat(parts_c).view(B, G), torch.cat(parts_ok).view(B, G)

    _PAD_CONST = {}

    @staticmethod
    def _padding_constants(device, G: int):
        """(G, 4) placeholder boxes [0, 0, 1, 1] (keep the CIoU arithmetic finite in padded slots), G zeros (class), G
        True / G False - made once per device and grown on demand (device-side fills: no upload)."""
        key = str(device)
        c = ObjectDetection._PAD_CONST.get(key)
        # (asked of CUDA tensors only: on a box without a HIP device the query itself raises, and on a GPU box a CPU call
        # would initialise the GPU as a side effect)
        capturing = torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()
        if c is None or c[0].shape[0] < G or capturing:
            n = max(G, 64)
            box = torch.zeros((n, 4), device=device)
            box[:, 2:] = 1.0
            c = (box, torch.zeros(n, device=device, dtype=torch.int64), torch.ones(n, device=device, dtype=torch.bool),
                 torch.zeros(n, device=device, dtype=torch.bool))
            if not capturing:  # constants made inside a capture live in the graph's pool
                ObjectDetection._PAD_CONST[key] = c
        return c

    @staticmethod
    def _match_padded(anchors: Tensor, gt: Tensor, col_ok: Tensor, topk: int) -> Tuple[Tensor, Tensor, Tensor]:
        """bbox_matching(relative=True) for the whole batch at once on padded ground truths: padded columns are
        masked out, so the result per image equals the per-image routine of the reference (:143-148, :252-284)
        while launching ~30 kernels instead of ~30 per image.  Also returns the per-(image, gt) top-k anchor
        indices (B, k, G)."""
        B, G = gt.shape[:2]
        A, device = anchors.shape[0], anchors.device
        if G == 0:
            return (torch.full((B, A), -1, device=device, dtype=torch.int64),
                    torch.zeros((B, A), device=device, dtype=torch.float32),
                    torch.zeros((B, topk, 0), device=device, dtype=torch.int64))
        ious = complete_box_iou(anchors, gt.reshape(B * G, 4)).reshape(A, B, G).permute(1, 0, 2).clamp(0)
        ious = ious * col_ok[:, None, :]
        top_v, top_i = torch.topk(ious, k=topk, dim=1)  # (B, k, G)
        in_topk = torch.zeros((B, A, G), dtype=torch.bool, device=device)
        in_topk.scatter_(1, top_i, True)
        in_topk &= col_ok[:, None, :]
        best_iou, best_gt = torch.max(ious * in_topk.float(), dim=2)  # (B, A)
        valid = in_topk.any(dim=2)
        assign = torch.where(valid, best_gt, torch.full_like(best_gt, -1))
        denom = torch.gather(top_v[:, 0, :], 1, best_gt)
        rel = (best_iou / denom).nan_to_num(0)
        return assign, torch.where(valid, rel, torch.zeros_like(rel)), top_i

    @classmethod
    def batched_matching(cls, anchors: Tensor, boxes: List[Tensor], topk: int) -> Tuple[Tensor, Tensor]:
        """(assignment, rel_iou), each (B, A), for per-image ground-truth lists."""
        gt, _, col_ok = cls._pad_targets(boxes, [b.new_zeros(b.shape[0], dtype=torch.int64) for b in boxes],
                                         anchors.device)
        return cls._match_padded(anchors, gt, col_ok, topk)[:2]

    @staticmethod
    def bbox_matching(anchors: Tensor, gt_boxes: Tensor, topk: int, relative: bool = False):
        """Top-k-per-GT one-to-many assignment (reference :252-284)."""
        A, G = anchors.shape[0], gt_boxes.shape[0]
        assign = torch.full((A,), -1, device=anchors.device)
        o2m = torch.zeros((A,), device=anchors.device)
        if G == 0:
            return assign, o2m
        ious = complete_box_iou(anchors, gt_boxes.to(anchors.dtype)).clamp(0)
        top_v, top_i = torch.topk(ious, k=topk, dim=0)
        in_topk = torch.zeros((A, G), dtype=torch.bool, device=anchors.device)
        in_topk.scatter_(0, top_i, True)
        best_iou, best_gt = torch.max(ious * in_topk.float(), dim=1)
        valid = in_topk.any(dim=1)
        assign = torch.where(valid, best_gt, assign)
        if not relative:
            return assign, torch.where(valid, best_iou, o2m)
        rel = (best_iou / top_v[0][best_gt]).nan_to_num(0)
        return assign, torch.where(valid, rel, o2m)

    def on_validation_start(self) -> None:
        from sihl_amd.metrics import BoxMeanAveragePrecision

        self._val_losses: List[Tensor] = []
        self.map_computer = BoxMeanAveragePrecision([1, min(self.max_instances, 10), self.max_instances])

    def validation_step(self, inputs, classes, boxes):
        """Reference :227-240: detections of ``forward`` go to the COCO-protocol mAP accumulator, the loss is the
        training loss on the same inputs."""
        _, scores, pred_classes, pred_boxes = self.forward(inputs)
        self.map_computer.update(
            [{"scores": s, "labels": c, "boxes": b} for s, c, b in zip(scores, pred_classes, pred_boxes)],
            [{"labels": c, "boxes": b} for c, b in zip(classes, boxes)])
        loss, metrics = self.training_step(inputs, classes, boxes, is_validating=True)
        self._val_losses.append(loss.detach())
        return loss, metrics

    def on_validation_end(self) -> Dict[str, float]:
        metrics = self.map_computer.compute() if hasattr(self, "map_computer") else {}
        metrics["loss"] = torch.stack(self._val_losses).mean().item() if self._val_losses else float("nan")
        return metrics
