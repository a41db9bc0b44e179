"""Test-infrastructure probe (GPU box; lives under tests/ because it runs the oracle): the reference's architecture and
training step as PLAIN PyTorch-ROCm on the same MI355X - the oracle model (a restatement of the reference's modules in
torch ops: MIOpen convs, ATen BatchNorm / loss, the reference's per-image matching loop and boolean-mask indexing with
their host syncs) on `cuda`, channels_last, bf16 autocast, AdamW + clip(0.1) - next to the sihl_amd step on the same
batch.  This is what "the reference on this hardware" means for BASELINE.json's metric; it is a baseline, not a checker.
usage: python tests/golden/torch_rocm_baseline_probe.py [batch] [steps]"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oracle  # noqa: E402
import sihl_amd  # noqa: E402
from sihl_amd.train import Trainer, configure_optimizer  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
images, targets = bench.synthetic_batch(bs, 512, dev, 0)


def timed(step, n, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# ---- plain PyTorch-ROCm: the oracle modules on the GPU
ns = types.SimpleNamespace(ResNetBackbone=oracle.ResNetBackbone, BiFPN=oracle.BiFPN, ObjectDetection=oracle.ObjectDetection,
                           SihlModel=oracle.SihlModel)
for amp in (torch.bfloat16, None):
    model = bench.build_model(ns, dev)
    model.train()
    opt = configure_optimizer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1)
    params = [p for p in model.parameters()]

    def torch_step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            feats = model.extract_features(images)
            loss = sum(h.training_step(feats, **t)[0] for h, t in zip(model.heads, targets))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()

    dt = timed(torch_step, steps)
    print(f"PyTorch-ROCm (oracle modules on cuda, channels_last, {'bf16 autocast' if amp else 'fp32'}): "
          f"{bs / dt:7.1f} img/s  {dt * 1e3:7.2f} ms/step", flush=True)
    del model, opt, params
    torch.cuda.empty_cache()

# ---- sihl_amd on the same batch
hip_ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                               ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = bench.build_model(hip_ns, dev)
tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16)
dt = timed(lambda: tr.step(images, targets), steps, warm=5)
print(f"sihl_amd (HIP kernels, bf16, two streams):                                  {bs / dt:7.1f} img/s  {dt * 1e3:7.2f} ms/step", flush=True)
