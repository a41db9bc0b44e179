"""Developer probe (GPU box): device time of the detection head's target matching (no network output involved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sihl_amd  # noqa: E402
from bench import synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
head = sihl_amd.heads.ObjectDetection([3, 64, 256] + [256] * 5, 80, 3, 7).to(dev)
images, targets = synthetic_batch(32, 512, dev, seed=0)
boxes, classes = targets[0]["boxes"], targets[0]["classes"]
level_hw = [(512 >> l, 512 >> l) for l in head.levels]
from sihl_amd import ops  # noqa: E402


def match():
    full = head._full_size(512, 512, dev)
    offsets, scales = ops.od_anchors(level_hw, dev)
    anchors = (offsets + scales) * full
    gt, gt_cls, col_ok = head._pad_targets(boxes, classes, dev)
    return head._match_padded(anchors, gt, col_ok, head.topk)


for _ in range(3):
    match()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    match()
e1.record()
torch.cuda.synchronize()
print(f"matching: {e0.elapsed_time(e1) / 20:.3f} ms per step (bs 32, 5456 anchors)")
