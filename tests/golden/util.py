"""Helpers to replay a golden case (tests/golden/cases.py) on some namespace of classes."""
import os
import re

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_npz(name):
    return dict(np.load(os.path.join(HERE, f"{name}.npz"), allow_pickle=False))


def _unflatten(prefix, data):
    if f"{prefix}.len" in data:
        return [_unflatten(f"{prefix}.{i}", data) for i in range(int(data[f"{prefix}.len"]))]
    return torch.from_numpy(data[prefix].copy())


def golden_inputs(data):
    keys = {re.match(r"in\.([^.]+)", k).group(1) for k in data if k.startswith("in.")}
    return {k: _unflatten(f"in.{k}", data) for k in keys}


def golden_state_dict(data, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in data.items() if k.startswith(prefix)}


def golden_results(data):
    return {k[4:]: torch.from_numpy(v.copy()) for k, v in data.items() if k.startswith("res.")}


def replay(case, ns, data, device="cpu", dtype=torch.float32, prepare=None):
    """Build the case's module from ``ns``, load the golden state_dict, run on ``device``."""
    torch.manual_seed(1234)
    m = case.build(ns)
    missing = m.load_state_dict(golden_state_dict(data), strict=True)
    m.train(case.train)
    m = m.to(device)
    if prepare is not None:
        m = prepare(m)

    def mv(x):
        if isinstance(x, list):
            return [mv(t) for t in x]
        return x.to(device=device, dtype=dtype) if x.is_floating_point() else x.to(device)

    inp = {k: mv(v) for k, v in golden_inputs(data).items()}
    res = case.run(m, inp)
    return m, {k: v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu() for k, v in res.items()}


def namespace_of(*modules):
    """Collect the classes of some modules into one attribute namespace (what a case's build() takes)."""
    class NS:
        pass

    for mod in modules:
        for k, v in vars(mod).items():
            if isinstance(v, type):
                setattr(NS, k, v)
    return NS


def quantized_copy(data):
    """Golden data with the inputs and every conv / linear weight rounded to bf16 (kept as fp32).

    Evaluating the fp32 oracle on this copy separates kernel error from operand quantisation: a
    bf16 run and this oracle see the same operands, so ReLU masks agree except where intermediate
    rounding differs."""
    out = dict(data)
    for k, v in data.items():
        is_in = k.startswith("in.") and v.dtype == np.float32
        is_w = k.startswith("sd.") and k.endswith("weight") and v.ndim >= 2
        if is_in or is_w:
            out[k] = torch.from_numpy(v.copy()).bfloat16().float().numpy()
    return out


class TinyBackbone(torch.nn.Module):
    """Plain-torch stand-in backbone honouring the reference's level contract (torchvision_backbone.py:161-186): returns
    [input, level 1 .. level 5], level l at 1 / 2^l of the input, ``out_channels`` lists their channels.  Used by the
    caller fixture (the real backbones are third-party and absent offline); conv -> BN -> ReLU stages so that the
    optimizer grouping sees backbone weights, biases-free convs and norm parameters."""

    def __init__(self, channels=(3, 8, 16, 24, 32, 48)):
        super().__init__()
        nn = torch.nn
        self.out_channels = list(channels)
        self.stages = nn.ModuleList(
            nn.Sequential(nn.Conv2d(cin, cout, 3, stride=2, padding=1, bias=(i == 0)), nn.BatchNorm2d(cout), nn.ReLU())
            for i, (cin, cout) in enumerate(zip(channels[:-1], channels[1:])))
        self.dummy_input = torch.zeros(1, 3, 64, 64)

    def forward(self, x):
        out = [x]
        for s in self.stages:
            out.append(s(out[-1]))
        return out


CALLER = dict(neck_channels=32, bottom=3, top=5, num_classes=6, image=64, batch=2,
              opt=dict(lr=1e-3, weight_decay=1e-2, backbone_lr_factor=0.1), warmup=3, t_max=10, sched_steps=8)


def caller_batch():
    g = torch.Generator().manual_seed(77)
    x = torch.rand(CALLER["batch"], 3, CALLER["image"], CALLER["image"], generator=g)
    boxes = [torch.tensor([[5.2, 7.9, 40.3, 33.1], [22.4, 30.6, 58.7, 61.2]]), torch.tensor([[10.5, 12.25, 30.75, 50.5]])]
    classes = [torch.tensor([1, 4]), torch.tensor([2])]
    return x, {"classes": classes, "boxes": boxes}


def emulate_bf16_storage(module):
    """Make a CPU fp32 oracle module keep its activations the way the bf16 HIP path does: every leaf module's output
    is rounded to bf16 (and carried on as fp32).  Autograd differentiates through the two casts, so the GRADIENT that
    flows back across the same boundary is rounded to bf16 too - fp32 accumulation inside each op, bf16 storage
    between ops, fp32 parameter gradients: the arithmetic model of the bf16 kernels.  The deviation of this run from
    the plain fp32 oracle on the same (bf16-rounded) operands is the noise floor any bf16 implementation lives with."""
    def hook(_m, _inp, out):
        if isinstance(out, torch.Tensor) and out.is_floating_point():
            return out.bfloat16().float()
        return out

    handles = [m.register_forward_hook(hook) for m in module.modules() if not any(True for _ in m.children())]
    return handles


def emulate_rounding_noise(module, seed):
    """Like emulate_bf16_storage, but every boundary value is perturbed by a random relative error of bf16's rounding
    size (uniform in +-2^-8, forward values only) instead of being rounded: ONE rounding pattern is one sample of the
    noise - gradients that are small differences of large sums (fusion weights, biases in front of a norm) move by very
    different amounts from pattern to pattern - so the floor of a tensor is taken as the largest deviation over the
    bf16 emulation and a few such samples."""
    g = torch.Generator().manual_seed(seed)

    def hook(_m, _inp, out):
        if isinstance(out, torch.Tensor) and out.is_floating_point():
            return out * (1 + (torch.rand(out.shape, generator=g) - 0.5) * 2.0 ** -7)
        return out

    return [m.register_forward_hook(hook) for m in module.modules() if not any(True for _ in m.children())]


def bf16_floor(case, ns, data, samples=3, envelope=False):
    """ref = fp32 oracle on `data`; floor[k] = max over the emulated runs of ||run[k] - ref[k]|| / ||ref[k]||.
    envelope=True: also returns env[k] = (lo, hi), the element-wise minimum / maximum over ref and the emulated runs."""
    _, ref = replay(case, ns, data)
    runs = [replay(case, ns, data, prepare=lambda m: (emulate_bf16_storage(m), m)[1])[1]]
    for s in range(samples):
        runs.append(replay(case, ns, data, prepare=lambda m, s=s: (emulate_rounding_noise(m, 100 + s), m)[1])[1])
    floor = {}
    for k, g in ref.items():
        if g.is_floating_point():
            n = float(g.float().norm().clamp(min=1e-30))
            floor[k] = max(float((r[k].float() - g.float()).norm()) / n for r in runs)
            # the same floor in worst-element terms (largest deviation over the tensor's largest magnitude)
            floor[k + "|max"] = max(float((r[k].float() - g.float()).abs().max()) for r in runs) / max(1e-6, float(g.abs().max()))
    if envelope:
        env = {}
        for k, g in ref.items():
            if g.is_floating_point():
                stack = torch.stack([g.float()] + [r[k].float() for r in runs])
                env[k] = (stack.min(0).values, stack.max(0).values)
        return ref, floor, env
    return ref, floor
