"""Developer probe (GPU box): time each phase of the flagship training step with progress output."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sihl_amd  # noqa: E402
from bench import build_model, synthetic_batch  # noqa: E402
import types  # noqa: E402


def log(*a):
    msg = " ".join(str(x) for x in a)
    print(msg, flush=True)
    open("gpurun_out/step_probe.log", "a").write(msg + "\n")


bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
dev = torch.device("cuda")
ns = types.SimpleNamespace(ResNetBackbone=sihl_amd.ResNetBackbone, BiFPN=sihl_amd.layers.BiFPN,
                           ObjectDetection=sihl_amd.heads.ObjectDetection, SihlModel=sihl_amd.SihlModel)
model = build_model(ns, dev)
images, targets = synthetic_batch(bs, 512, dev, 0)
model.train()


def T():
    torch.cuda.synchronize()
    return time.time()


for it in range(3):
    t0 = T()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
        levels = model.backbone(images)
    levels = [t if i == 0 else t.to(dt) for i, t in enumerate(levels)]
    t1 = T(); log(f"it{it} backbone fwd {t1-t0:.3f}")
    feats = model.neck(levels)
    t2 = T(); log(f"it{it} neck fwd {t2-t1:.3f}")
    loss, m = model.heads[0].training_step(feats, **targets[0])
    t3 = T(); log(f"it{it} head train fwd {t3-t2:.3f} loss {float(loss):.4f}")
    loss.backward()
    t4 = T(); log(f"it{it} backward {t4-t3:.3f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    model.zero_grad(set_to_none=True)
model.eval()
with torch.no_grad():
    for it in range(3):
        t0 = T()
        feats = model.neck(levels)
        t1 = T()
        out = model.heads[0](feats)
        t2 = T(); log(f"eval it{it} neck {t1-t0:.4f} head fwd {t2-t1:.4f}")
