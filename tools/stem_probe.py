"""Developer probe (GPU box): the native stem path and the ATen stem path (both bf16) against an fp32 run of the same module."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sihl_amd  # noqa: E402

torch.manual_seed(5)
model = sihl_amd.ResNetBackbone("resnet18", native=True, top_level=2).cuda().train()
x = torch.rand(4, 3, 96, 128, device="cuda")
state = {k: v.clone() for k, v in model.state_dict().items()}


def run(mode):
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    os.environ.pop("SIHL_ATEN_STEM", None)
    if mode != "native":
        os.environ["SIHL_ATEN_STEM"] = "1"
    if mode == "fp32":
        outs = model(x)
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs = model(x)
    gc = torch.Generator(device="cuda").manual_seed(9)
    sum((o.float() * torch.randn(o.shape, device="cuda", generator=gc)).mean() for o in outs[1:]).backward()
    t = model.model
    return [o.float().detach() for o in outs[1:]] + [t.conv1.weight.grad.clone(), t.bn1.weight.grad.clone(), t.bn1.bias.grad.clone()]


ref = run("fp32")
for mode in ("native", "aten", "native", "aten"):
    got = run(mode)
    print(mode, " ".join(f"{float((a - b).norm() / b.norm()):.3e}" for a, b in zip(got, ref)))
