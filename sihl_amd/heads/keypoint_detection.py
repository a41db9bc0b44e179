"""KeypointDetection head on the HIP path (reference src/sihl/heads/keypoint_detection.py:15-378, FCPose-style
instance-aware heatmaps).

Same front end as the detection / instance-segmentation heads, on the HIP kernels: 1x1 conv+BN laterals into the flat
(B, P, C) buffer, location MLP over all positions, per-image top-K, gather, presence and kernel MLPs on the K rows, mask
branch (1x1 conv+BN, 3x3 conv+BN+SiLU to 32 channels).  The per-instance 34 -> 32 -> 32 -> num_keypoints network and
the heatmap argmax / losses are device tensor ops in fp32 (no dedicated kernel yet: unlike the mask decode there is no
full-resolution output to write, the heatmaps stay at the mask level).
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


def dynamic_keypoint_net(feats: Tensor, weights: Tensor, c: int, k: int) -> Tensor:
    """feats (n, c+2, h, w), weights (n, (c+2)c + c + cc + c + ck + k) -> heatmap logits (n, k, h, w)."""
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


class KeypointDetection(nn.Module):
    def __init__(self, in_channels: List[int], num_keypoints: int, mask_level: int = 3, bottom_level: int = 5,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 4, max_instances: int = 100) -> None:
        assert num_keypoints > 0, num_keypoints
        assert len(in_channels) > top_level, (len(in_channels), top_level)
        assert 0 < bottom_level <= top_level, (bottom_level, top_level)
        assert num_channels % 4 == 0, num_channels
        assert num_layers >= 0, num_layers
        assert max_instances > 0, max_instances
        super().__init__()
        self.in_channels, self.num_keypoints, self.mask_level = in_channels, num_keypoints, mask_level
        self.bottom_level, self.top_level = bottom_level, top_level
        self.levels = range(bottom_level, top_level + 1)
        self.num_channels, self.num_layers = num_channels, num_layers
        self.max_instances, self.topk = max_instances, 9
        mlp = partial(MLP, norm_layer=nn.LayerNorm, activation_layer=nn.SiLU)
        self.laterals = nn.ModuleList(
            [Conv2dNormActivation(in_channels[l], num_channels, 1, activation_layer=None) for l in self.levels])
        hidden = [num_channels] * num_layers
        self.loc_head = mlp(num_channels, hidden + [1])
        self.loc_head[-2].bias.data.fill_(-5.0)  # reference :69
        self.presence_head = mlp(num_channels, hidden + [num_keypoints])
        c = self.mask_num_channels = 32
        self.kernel_head = mlp(num_channels, hidden + [(c + 2) * c + c + c * c + c + c * num_keypoints + num_keypoints])
        self.mask_lateral = Conv2dNormActivation(in_channels[mask_level], num_channels, 1, activation_layer=None)
        self.mask_head = Conv2dNormActivation(num_channels, c, 3, activation_layer=nn.SiLU)
        self.output_shapes = {"num_instances": ("batch_size",), "scores": ("batch_size", max_instances),
                              "presence": ("batch_size", max_instances, num_keypoints),
                              "keypoints": ("batch_size", max_instances, num_keypoints, 2)}

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def keypoints_to_boxes(keypoints: Tensor, presence: Tensor) -> Tensor:
        assert presence.dtype == torch.bool
        lo = keypoints.masked_fill(~presence[..., None], float("inf")).amin(dim=1)
        hi = keypoints.masked_fill(~presence[..., None], float("-inf")).amax(dim=1)
        return torch.cat([lo, hi], dim=-1)

    @staticmethod
    def keypoints_to_heatmaps(keypoints: Tensor, presence: Tensor, height: int, width: int, img_height: int,
                              img_width: int) -> Tensor:
        xs = (keypoints[:, :, 0] * ((width - 1) / (img_width - 1))).clamp(0, width - 1).round().to(torch.int64)
        ys = (keypoints[:, :, 1] * ((height - 1) / (img_height - 1))).clamp(0, height - 1).round().to(torch.int64)
        gx, gy = F.one_hot(xs, width).to(torch.float32), F.one_hot(ys, height).to(torch.float32)
        return presence[:, :, None, None] * gx.unsqueeze(2) * gy.unsqueeze(3)

    def _level_hw(self, inputs: List[Tensor]) -> List[Tuple[int, int]]:
        return [tuple(inputs[l].shape[2:]) for l in self.levels]

    def _flat_feats(self, inputs: List[Tensor]) -> Tensor:
        feats = [lat.forward_nhwc(ops.nhwc(inputs[l])) for l, lat in zip(self.levels, self.laterals)]
        B, C = feats[0].shape[0], feats[0].shape[-1]
        return torch.cat([f.reshape(B, -1, C) for f in feats], dim=1)

    def _mask_feats(self, inputs: List[Tensor]) -> Tensor:
        x = self.mask_head.forward_nhwc(self.mask_lateral.forward_nhwc(ops.nhwc(inputs[self.mask_level])))
        return ops.nchw_view(x).float()  # (B, c, h, w)

    @staticmethod
    def _grid(h: int, w: int, device) -> Tensor:
        ys = (torch.arange(h, device=device, dtype=torch.float32) + 0.5) / h
        xs = (torch.arange(w, device=device, dtype=torch.float32) + 0.5) / w
        return torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)])  # (2, h, w)

    def get_saliency(self, inputs: List[Tensor]) -> Tensor:
        return self.forward(inputs, output_heatmaps=True).amax(dim=(1, 2))

    # ------------------------------------------------------------------ inference
    def forward(self, inputs: List[Tensor], output_heatmaps: bool = False):
        B, _, H, W = inputs[0].shape
        device = inputs[self.bottom_level].device
        c, nk, K = self.mask_num_channels, self.num_keypoints, self.max_instances
        flat = self._flat_feats(inputs)
        P = flat.shape[1]
        loc_logits = self.loc_head(flat.view(B * P, -1))
        top_vals, top_idx = ops.topk_rows(loc_logits, B, P, K, estride=loc_logits.stride(0))
        sel = ops.gather_rows(flat, top_idx).view(B * K, -1)
        scores = top_vals.sigmoid()
        num_instances = (scores > 0.5).sum(dim=1)
        mask_feats = self._mask_feats(inputs)
        h, w = mask_feats.shape[2:]
        offsets, _ = ops.od_anchors(self._level_hw(inputs), device)
        centres = offsets[top_idx.long().reshape(-1), :2]  # (B*K, 2)
        rel = self._grid(h, w, device)[None] - centres[:, :, None, None]
        feats = torch.cat([mask_feats[:, None].expand(B, K, c, h, w).reshape(B * K, c, h, w), rel], dim=1)
        kernels, presence = forward_many([self.kernel_head, self.presence_head], sel)  # one launch in inference
        heat = dynamic_keypoint_net(feats, kernels.float(), c, nk).reshape(B, K, nk, h, w)
        presence = presence.reshape(B, K, nk).float().sigmoid()
        if output_heatmaps:
            return heat.flatten(3, 4).softmax(3).reshape(heat.shape)
        flat_idx = heat.flatten(3, 4).max(3).indices
        ky, kx = flat_idx // h, flat_idx % h  # the reference uses the mask HEIGHT for both (:165); kept for parity
        ky = (ky.to(torch.float32) + 0.5) / h * H
        kx = (kx.to(torch.float32) + 0.5) / w * W
        return num_instances, scores, presence, torch.stack([kx, ky], dim=3)

    # ------------------------------------------------------------------ training
    def training_step(self, inputs: List[Tensor], presence: List[Tensor], keypoints: List[Tensor],
                      is_validating: bool = False) -> Tuple[Tensor, Dict[str, float]]:
        assert len(inputs) > self.top_level, "too few input levels"
        device = inputs[self.bottom_level].device
        B, _, H, W = inputs[0].shape
        c, nk = self.mask_num_channels, self.num_keypoints
        presence = [p.to(device) for p in presence]
        keypoints = [k.to(device) for k in keypoints]
        keep = [p.any(dim=1) for p in presence]  # instances without a visible keypoint are dropped (:186-192)
        keypoints = [k[m] for k, m in zip(keypoints, keep)]
        presence = [p[m] for p, m in zip(presence, keep)]
        boxes = [self.keypoints_to_boxes(k, p) for k, p in zip(keypoints, presence)]
        offsets, scales = ops.od_anchors(self._level_hw(inputs), device)
        anchors = (offsets + scales) * torch.tensor([[W, H, W, H]], device=device, dtype=torch.float32)
        assignment, rel_iou = ObjectDetection.batched_matching(anchors, boxes, self.topk)

        flat = self._flat_feats(inputs)
        P, C = flat.shape[1], flat.shape[2]
        loc_logits = self.loc_head(flat.view(B * P, C)).reshape(B, P)
        loc_target = (rel_iou == 1.0).to(torch.float32)
        loc_loss = F.binary_cross_entropy_with_logits(loc_logits.float(), loc_target, reduction="none")
        loc_loss = loc_loss.sum() / loc_target.sum()
        z = torch.zeros_like(loc_loss)
        if rel_iou.max() == 0:
            return loc_loss, {"location_loss": loc_loss, "keypoint_loss": z, "presence_loss": z}

        o2m = rel_iou > 0
        wts = rel_iou[o2m].reshape(-1, 1)
        pos = o2m.nonzero()  # (n, 2): image, position
        sel = flat[o2m]
        counts = torch.tensor([0] + [p.shape[0] for p in presence[:-1]], device=device).cumsum(0)
        flat_gt = counts[pos[:, 0]] + assignment[o2m]
        target_presence = torch.cat(presence)[flat_gt]
        presence_loss = F.binary_cross_entropy_with_logits(self.presence_head(sel).float(),
                                                           target_presence.to(torch.float32), reduction="none")
        presence_loss = (wts * presence_loss).sum() / wts.sum()

        mask_feats = self._mask_feats(inputs)
        h, w = mask_feats.shape[2:]
        rel = self._grid(h, w, device)[None] - offsets[pos[:, 1], :2][:, :, None, None]
        feats = torch.cat([mask_feats[pos[:, 0]], rel], dim=1)
        heat = dynamic_keypoint_net(feats, self.kernel_head(sel).float(), c, nk)
        target_heat = self.keypoints_to_heatmaps(torch.cat(keypoints)[flat_gt], target_presence, h, w, H, W)
        kp_loss = F.cross_entropy(heat.flatten(2).transpose(1, 2), target_heat.flatten(2).transpose(1, 2),
                                  reduction="none")  # classes = the h*w cells
        kp_loss = (wts * kp_loss).sum() / wts.sum()
        loss = loc_loss + kp_loss + presence_loss
        return loss, {"location_loss": loc_loss, "keypoint_loss": kp_loss, "presence_loss": presence_loss}

    def on_validation_start(self) -> None:
        from sihl_amd.metrics import PercentageOfCorrectKeypoints

        self._val_losses: List[Tensor] = []
        self.pck_computer = PercentageOfCorrectKeypoints(threshold=0.05)

    def validation_step(self, inputs, keypoints, presence):
        """Reference :322-341: PCK@0.05 of ``forward``'s instances (keypoints as image fractions, the first num_instances of
        every image) against the ground truth, plus the training loss on the same inputs."""
        B, _, H, W = inputs[0].shape
        with torch.no_grad():
            num_instances, _, keypoint_scores, pred_keypoints = self.forward(inputs)
        full = torch.tensor([[[W, H]]], device=pred_keypoints.device, dtype=torch.float32)
        counts = num_instances.tolist()  # one host read per validation step (the metric pairs instances on the host anyway)
        for b in range(B):
            k = int(counts[b])
            self.pck_computer.update(pred_keypoints[b, :k].float() / full, keypoint_scores[b, :k],
                                     keypoints[b].to(full.device).float() / full, presence[b])
        loss, metrics = self.training_step(inputs, keypoints=keypoints, presence=presence, is_validating=True)
        self._val_losses.append(loss.detach())
        return loss, metrics

    def on_validation_end(self) -> Dict[str, float]:
        metrics = self.pck_computer.compute() if hasattr(self, "pck_computer") else {}
        metrics["loss"] = torch.nanmean(torch.stack(self._val_losses).float()).item() if self._val_losses else float("nan")
        return metrics
