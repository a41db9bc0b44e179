#!/usr/bin/env python3
"""PMC target: a few launches of the dominant kernels on the flagship L3 shape (bs 32, 64x64, 256->256, 3x3, bf16)
so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be read per dispatch (run once per counter)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sihl_amd import ops  # noqa: E402

dev, dt = "cuda", torch.bfloat16
x = torch.randn(32, 64, 64, 256, device=dev, dtype=dt)
w = torch.randn(256, 3, 3, 256, device=dev, dtype=dt) * 0.05
dy = torch.randn(32, 64, 64, 256, device=dev, dtype=dt)
for _ in range(5):
    ops.conv2d_raw(x, w, None, 1, 1, 1, act="relu", stats_mode=2)
    ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
torch.cuda.synchronize()
