"""DepthEstimation head on the HIP path (reference src/sihl/heads/depth_estimation.py:15-122; AdaBins on the
SemanticSegmentation decoder - SURVEY 8f rank 4).

The dense work - SPPM, laterals, UAFM fusions, conv towers of both the per-pixel bin scores and the per-image bin
widths - is the segmentation head's HIP path; the bin arithmetic (normalised widths -> centres, score-weighted depth,
scale-invariant log loss, chamfer loss between bin centres and target depths) is fp32 device arithmetic as in the
reference.
"""
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

from sihl_amd import ops
from sihl_amd.heads.semantic_segmentation import SemanticSegmentation, _PlainConv
from sihl_amd.layers.convblocks import SequentialConvBlocks

EPS = 1e-5  # sihl.utils.EPS


class DepthEstimation(SemanticSegmentation):
    def __init__(self, in_channels: List[int], lower_bound: float, upper_bound: float, bottom_level: int = 3,
                 top_level: int = 5, num_channels: int = 256, num_layers: int = 1, num_bins: int = 256) -> None:
        assert lower_bound < upper_bound
        assert len(in_channels) > top_level >= bottom_level > 0
        assert num_channels > 0 and num_layers > 0
        assert num_bins > 1
        super().__init__(in_channels=in_channels, num_classes=num_bins, num_channels=num_channels,
                         bottom_level=bottom_level, top_level=top_level, num_layers=num_layers)
        self.num_bins = num_bins
        self.lower_bound, self.upper_bound = lower_bound, upper_bound
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
        x = self.bin_head[0].forward_nhwc(ops.nhwc(inputs[self.top_level]))
        x = _PlainConv(self.bin_head[1]).forward_nhwc(x)  # (B, h, w, bins)
        widths = x.float().mean(dim=(1, 2)).relu() + EPS  # AdaptiveAvgPool2d(1) + Flatten
        widths = widths / widths.sum(dim=1, keepdim=True)
        return widths.cumsum(dim=1) - widths / 2

    def get_depth_map(self, inputs: List[Tensor], bin_centers: Tensor) -> Tensor:
        weights = self._logits_nhwc(inputs).float().relu() + EPS  # (B, h, w, bins)
        weights = weights / weights.sum(dim=3, keepdim=True)
        depth = (weights * bin_centers[:, None, None, :]).sum(dim=3)
        return depth.clamp(0, 1)[:, None]  # (B, 1, h, w)

    def forward(self, inputs: List[Tensor]) -> Tensor:
        depth = self.denormalize(self.get_depth_map(inputs, self.get_bin_centers(inputs)))
        return F.interpolate(depth, size=inputs[0].shape[2:]).squeeze(1)

    def training_step(self, inputs: List[Tensor], targets: Tensor, masks: Tensor) -> Tuple[Tensor, Dict[str, float]]:
        device = inputs[self.top_level].device
        targets, masks = targets.to(device), masks.to(device)
        B = targets.shape[0]
        masks, targets = masks[:, None], self.normalize(targets[:, None].float())
        centers = self.get_bin_centers(inputs)
        depth = self.get_depth_map(inputs, centers)
        pred_shape = depth.shape[2:]
        depth = F.interpolate(depth, size=targets.shape[2:])
        g = (depth[masks] + EPS).log() - (targets[masks] + EPS).log()
        gm = g.mean()
        pix_loss = torch.sqrt(g.var() + 0.15 * gm * gm) * 10
        masks = F.interpolate(masks.to(torch.uint8), size=pred_shape, mode="nearest") > 0
        targets = F.interpolate(targets, size=pred_shape)
        hist = []
        for b in range(B):
            dist = centers[b][None, :] - targets[b][masks[b]][:, None]
            dist = dist * dist
            hist.append(dist.min(dim=1).values.mean() + dist.min(dim=0).values.mean())
        hist_loss = torch.stack(hist).mean()
        return pix_loss + hist_loss, {"pixel_loss": pix_loss, "hist_loss": hist_loss}

    def on_validation_start(self) -> None:
        """The reference keeps a MeanMetric of the loss and torchmetrics' MeanAbsoluteError / MeanSquaredError(squared=False)
        over the masked pixels (depth_estimation.py:124-127); here: four device-side accumulators, read once at the end."""
        self._val_losses: List[Tensor] = []
        self._val_acc = None  # [sum |err|, sum err^2, count] over every masked pixel seen, float64 on the device

    def validation_step(self, inputs: List[Tensor], targets: Tensor, masks: Tensor):
        with torch.no_grad():
            loss, _ = self.training_step(inputs, targets, masks)
            depth = self.forward(inputs)  # (B, H, W), denormalised
            m = masks.to(depth.device)
            err = (depth - targets.to(depth.device).float())[m].double()
            acc = torch.stack([err.abs().sum(), (err * err).sum(), m.sum().double()])
            self._val_acc = acc if self._val_acc is None else self._val_acc + acc
        self._val_losses.append(loss.detach())
        return loss, {}

    def on_validation_end(self) -> Dict[str, float]:
        """{"loss", "rmse", "mae"} as the reference (depth_estimation.py:141-146); NaN losses are ignored in the mean like
        MeanMetric(nan_strategy="ignore")."""
        out = {"loss": float("nan"), "rmse": float("nan"), "mae": float("nan")}
        if self._val_losses:
            out["loss"] = torch.nanmean(torch.stack(self._val_losses).float()).item()
        if self._val_acc is not None:
            abs_sum, sq_sum, count = self._val_acc.tolist()
            if count > 0:
                out["mae"], out["rmse"] = abs_sum / count, (sq_sum / count) ** 0.5
        return out
