"""COCO-protocol mean average precision for boxes and masks, PCK, segmentation confusion counts (box mAP: the metric the reference's ObjectDetection validation reports through
torchmetrics' MeanAveragePrecision with the faster_coco_eval backend, src/sihl/heads/object_detection.py:219-250).

Neither torchmetrics nor a COCO evaluator is available offline, so the published COCO detection-evaluation protocol is
restated here (host-side numpy; validation is not on the hot path): IoU thresholds 0.50:0.05:0.95, greedy matching of
score-sorted detections to the best still-unmatched ground truth of the same class and image, 101-point interpolated
precision, area ranges all / small (< 32^2) / medium / large (>= 96^2), maxDets from ``max_detection_thresholds``.
"parity unpinned": there is no reference output to compare with here; tests/test_host_cpu.py checks hand-computed cases.
"""
from typing import Dict, List, Sequence

import numpy as np
import torch

IOU_THRESHOLDS = np.linspace(0.5, 0.95, 10)
RECALL_THRESHOLDS = np.linspace(0.0, 1.0, 101)
AREA_RANGES = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}


def _iou_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(n, 4) x (m, 4) xyxy boxes -> (n, m) IoU."""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = np.clip(rb - lt, 0, None).prod(-1)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(area_a[:, None] + area_b[None, :] - inter, 1e-12)


class BoxMeanAveragePrecision:
    def __init__(self, max_detection_thresholds: Sequence[int] = (1, 10, 100)) -> None:
        self.max_dets = sorted(int(m) for m in max_detection_thresholds)
        self.reset()

    def reset(self) -> None:
        self._images: List[Dict[str, np.ndarray]] = []

    def update(self, preds: List[Dict[str, torch.Tensor]], targets: List[Dict[str, torch.Tensor]]) -> None:
        """preds: per image {"scores" (n,), "labels" (n,), "boxes" (n, 4)}; targets: {"labels" (g,), "boxes" (g, 4)}."""
        def np_(t):
            return t.detach().float().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64)
        for p, t in zip(preds, targets):
            det, gt = np_(p["boxes"]).reshape(-1, 4), np_(t["boxes"]).reshape(-1, 4)
            self._store(np_(p["scores"]), np_(p["labels"]), np_(t["labels"]), _iou_matrix(det, gt),
                        (det[:, 2] - det[:, 0]) * (det[:, 3] - det[:, 1]), (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1]))

    def _store(self, scores, labels, gt_labels, ious, det_area, gt_area) -> None:
        """One image: detection scores / labels (n,), ground-truth labels (g,), the (n, g) IoU matrix and both area vectors -
        everything the COCO protocol needs, whatever the geometry (boxes here, masks in MaskMeanAveragePrecision)."""
        self._images.append({"scores": np.asarray(scores, dtype=np.float64).reshape(-1),
                             "labels": np.asarray(labels).reshape(-1).astype(np.int64),
                             "gt_labels": np.asarray(gt_labels).reshape(-1).astype(np.int64),
                             "ious": np.asarray(ious, dtype=np.float64).reshape(np.asarray(scores).size, np.asarray(gt_labels).size),
                             "det_area": np.asarray(det_area, dtype=np.float64).reshape(-1),
                             "gt_area": np.asarray(gt_area, dtype=np.float64).reshape(-1)})

    # ------------------------------------------------------------------ COCO evaluate + accumulate
    def _evaluate_image(self, img, cls: int, area, max_det: int):
        g_idx = np.nonzero(img["gt_labels"] == cls)[0]
        d_idx = np.nonzero(img["labels"] == cls)[0]
        score = img["scores"][d_idx]
        order = np.argsort(-score, kind="mergesort")[:max_det]
        d_idx, score = d_idx[order], score[order]
        g_area = img["gt_area"][g_idx]
        g_ignore = (g_area < area[0]) | (g_area > area[1])
        g_order = np.argsort(g_ignore, kind="mergesort")  # evaluated ground truths first
        g_idx, g_ignore = g_idx[g_order], g_ignore[g_order]
        ious = img["ious"][np.ix_(d_idx, g_idx)] if len(d_idx) and len(g_idx) else np.zeros((len(d_idx), len(g_idx)))
        T, D, G = len(IOU_THRESHOLDS), len(d_idx), len(g_idx)
        d_match = np.zeros((T, D), dtype=bool)
        d_ignore = np.zeros((T, D), dtype=bool)
        for ti, thr in enumerate(IOU_THRESHOLDS):
            g_taken = np.zeros(G, dtype=bool)
            for di in range(D):
                best, m = min(thr, 1 - 1e-10), -1
                for gi in range(G):
                    if g_taken[gi]:
                        continue
                    if m > -1 and not g_ignore[m] and g_ignore[gi]:
                        break  # already matched to an evaluated ground truth: do not trade it for an ignored one
                    if ious[di, gi] < best:
                        continue
                    best, m = ious[di, gi], gi
                if m > -1:
                    g_taken[m] = True
                    d_match[ti, di] = True
                    d_ignore[ti, di] = g_ignore[m]
        d_area = img["det_area"][d_idx]
        outside = (d_area < area[0]) | (d_area > area[1])
        d_ignore = d_ignore | (~d_match & outside[None, :])  # unmatched detections outside the range do not count
        return score, d_match, d_ignore, int((~g_ignore).sum())

    def _accumulate(self, area_name: str, max_det: int):
        classes = sorted({int(c) for img in self._images for c in np.concatenate([img["labels"], img["gt_labels"]])})
        T, R = len(IOU_THRESHOLDS), len(RECALL_THRESHOLDS)
        precision = -np.ones((T, R, len(classes)))
        recall = -np.ones((T, len(classes)))
        for ci, cls in enumerate(classes):
            evals = [self._evaluate_image(img, cls, AREA_RANGES[area_name], max_det) for img in self._images]
            n_gt = sum(e[3] for e in evals)
            if n_gt == 0:
                continue
            scores = np.concatenate([e[0] for e in evals])
            order = np.argsort(-scores, kind="mergesort")
            match = np.concatenate([e[1] for e in evals], axis=1)[:, order]
            ignore = np.concatenate([e[2] for e in evals], axis=1)[:, order]
            tps = np.cumsum(match & ~ignore, axis=1).astype(np.float64)
            fps = np.cumsum(~match & ~ignore, axis=1).astype(np.float64)
            for ti in range(T):
                tp, fp = tps[ti], fps[ti]
                rc = tp / n_gt
                pr = tp / np.maximum(tp + fp, np.spacing(1))
                recall[ti, ci] = rc[-1] if len(rc) else 0.0
                for i in range(len(pr) - 1, 0, -1):  # precision envelope
                    pr[i - 1] = max(pr[i - 1], pr[i])
                idx = np.searchsorted(rc, RECALL_THRESHOLDS, side="left")
                q = np.zeros(R)
                ok = idx < len(pr)
                q[ok] = pr[idx[ok]]
                precision[ti, :, ci] = q
        return precision, recall

    def compute(self) -> Dict[str, float]:
        def mean_valid(x):
            x = x[x > -1]
            return float(x.mean()) if x.size else -1.0
        out: Dict[str, float] = {}
        top = self.max_dets[-1]
        p_all, _ = self._accumulate("all", top)
        out["map"] = mean_valid(p_all)
        out["map_50"] = mean_valid(p_all[0])
        out["map_75"] = mean_valid(p_all[5])
        for name in ("small", "medium", "large"):
            p, r = self._accumulate(name, top)
            out[f"map_{name}"] = mean_valid(p)
            out[f"mar_{name}"] = mean_valid(r)
        for m in self.max_dets:
            _, r = self._accumulate("all", m)
            out[f"mar_{m}"] = mean_valid(r)
        return out


class MaskMeanAveragePrecision(BoxMeanAveragePrecision):
    """The same COCO protocol with ``iou_type="segm"`` (the reference's InstanceSegmentation validation,
    src/sihl/heads/instance_segmentation.py:299-330): IoU between boolean masks, areas = mask pixel counts.  The (n, g) IoU
    matrix of an image is computed ON THE DEVICE at update time - one matrix product of the flattened masks - so the masks
    themselves (100 x H x W per image) never travel to the host.  "parity unpinned" like the box metric."""

    def update(self, preds: List[Dict[str, torch.Tensor]], targets: List[Dict[str, torch.Tensor]]) -> None:
        """preds: per image {"scores" (n,), "labels" (n,), "masks" (n, H, W) bool}; targets: {"labels" (g,), "masks" (g, H, W)}."""
        for p, t in zip(preds, targets):
            dm, gm = p["masks"], t["masks"]
            dm = dm.reshape(dm.shape[0], dm[0].numel() if dm.shape[0] else 0).to(torch.float32)
            gm = gm.reshape(gm.shape[0], gm[0].numel() if gm.shape[0] else 0).to(device=dm.device, dtype=torch.float32)
            d_area, g_area = dm.sum(1).double(), gm.sum(1).double()
            inter = (dm @ gm.t()).double() if dm.shape[0] and gm.shape[0] else torch.zeros((dm.shape[0], gm.shape[0]), dtype=torch.float64)
            union = (d_area[:, None] + g_area[None, :] - inter).clamp(min=1e-12) if inter.numel() else inter
            ious = inter / union if inter.numel() else inter
            self._store(p["scores"].detach().float().cpu().numpy(), p["labels"].detach().cpu().numpy(),
                        t["labels"].detach().cpu().numpy(), ious.cpu().numpy(), d_area.cpu().numpy(), g_area.cpu().numpy())


class PercentageOfCorrectKeypoints:
    """PCK as the reference's KeypointDetection validation computes it (src/sihl/utils/pck.py:8-181, a torchmetrics
    ``Metric`` there): per image, predictions and ground truths are paired GREEDILY by the mean distance over the keypoints
    visible in both (lowest cost first, each used once; pairs without a mutually visible keypoint never match); for every
    pair, the ground truth's visible keypoints count as correct when the prediction lies within ``threshold`` (coordinates
    are image fractions); visible keypoints of unmatched ground truths count as missed.  The two counters live on the
    device; the cost matrix is one broadcast, the greedy loop runs on its (n, g) host copy."""

    def __init__(self, threshold: float = 0.05) -> None:
        self.threshold = float(threshold)
        self.reset()

    def reset(self) -> None:
        self.correct, self.total = 0, 0

    def update(self, pred_keypoints: torch.Tensor, pred_presence: torch.Tensor, gt_keypoints: torch.Tensor,
               gt_presence: torch.Tensor) -> None:
        """(n, K, 2), (n, K), (g, K, 2), (g, K)."""
        n, g = pred_keypoints.shape[0], gt_keypoints.shape[0]
        gt_vis = gt_presence.to(pred_keypoints.device) > 0
        if n == 0 or g == 0:
            self.total += int(gt_vis.sum()) if g else 0
            return
        gt_keypoints = gt_keypoints.to(pred_keypoints.device)
        dist = (pred_keypoints[:, None].float() - gt_keypoints[None].float()).norm(dim=-1)  # (n, g, K)
        mutual = (pred_presence > 0)[:, None, :] & gt_vis[None, :, :]
        cnt = mutual.sum(-1)
        cost = torch.where(cnt > 0, (dist * mutual).sum(-1) / cnt.clamp(min=1), torch.full_like(dist[..., 0], float("inf")))
        cost_h = cost.double().cpu().numpy().copy()
        within = ((dist <= self.threshold) & gt_vis[None]).sum(-1).cpu().numpy()  # correct keypoints of every (pred, gt) pair
        vis_h = gt_vis.sum(-1).cpu().numpy()
        matched = np.zeros(g, dtype=bool)
        while True:
            flat = int(np.argmin(cost_h))  # first minimum in row-major order, as `nonzero()[0]` in the reference
            i, j = divmod(flat, g)
            if not np.isfinite(cost_h[i, j]):
                break
            matched[j] = True
            if vis_h[j] > 0:
                self.correct += int(within[i, j])
                self.total += int(vis_h[j])
            cost_h[i, :] = np.inf
            cost_h[:, j] = np.inf
        self.total += int(vis_h[~matched].sum())

    def compute(self) -> Dict[str, float]:
        return {"PCK": (self.correct / self.total) if self.total else 0.0}


class SegmentationConfusion:
    """Pixel accuracy and mean IoU of the reference's SemanticSegmentation validation (src/sihl/heads/
    semantic_segmentation.py:94-120: torchmetrics ``Accuracy`` and ``JaccardIndex``, task "multiclass", default averaging),
    restated from their published definitions on a confusion matrix that is accumulated ON THE DEVICE (one bincount per
    validation step, no host synchronisation) and read once at the end:
      * pixels whose target is ``ignore_index`` are dropped;
      * pixel accuracy = micro average = correctly classified pixels / counted pixels;
      * mean IoU = macro average of TP / (TP + FP + FN) over the classes that occur in the targets or the predictions
        (a class absent from both has no IoU and does not enter the mean); when ``ignore_index`` is itself a class index it
        is excluded from the mean as well.
    "parity unpinned": torchmetrics is not importable here; tests check hand-computed cases."""

    def __init__(self, num_classes: int, ignore_index=None) -> None:
        self.num_classes, self.ignore_index = int(num_classes), ignore_index
        self.confmat = None

    def update(self, pred_classes: torch.Tensor, targets: torch.Tensor) -> None:
        """pred_classes, targets: integer tensors of the same shape (any device); rows = target, columns = prediction."""
        t = targets.reshape(-1).to(torch.int64)
        p = pred_classes.reshape(-1).to(device=t.device, dtype=torch.int64)
        keep = (t >= 0) & (t < self.num_classes)
        if self.ignore_index is not None:
            keep &= t != self.ignore_index
        idx = torch.where(keep, t * self.num_classes + p.clamp(0, self.num_classes - 1), torch.full_like(t, self.num_classes ** 2))
        counts = torch.bincount(idx, minlength=self.num_classes ** 2 + 1)[:-1]
        self.confmat = counts if self.confmat is None else self.confmat + counts

    def compute(self) -> Dict[str, float]:
        if self.confmat is None:
            return {"pixel_accuracy": float("nan"), "mean_iou": float("nan")}
        cm = self.confmat.reshape(self.num_classes, self.num_classes).double().cpu()
        tp = cm.diag()
        total = cm.sum()
        denom = cm.sum(0) + cm.sum(1) - tp
        present = denom > 0
        if self.ignore_index is not None and 0 <= self.ignore_index < self.num_classes:
            present[self.ignore_index] = False
        iou = tp[present] / denom[present]
        return {"pixel_accuracy": float(tp.sum() / total) if total > 0 else float("nan"),
                "mean_iou": float(iou.mean()) if present.any() else float("nan")}
