"""GPU: the native (HIP) residual stages of the ResNet backbone against the CPU oracle ResNet (plain torch.nn)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,size", [("resnet50", 128), ("resnet18", 64), ("resnet50", 96)])
@pytest.mark.parametrize("train", [True, False])
def test_native_backbone_fp32_matches_oracle(name, size, train):
    import oracle
    import sihl_amd

    torch.manual_seed(0)
    ref = oracle.ResNetBackbone(name)
    hip = sihl_amd.ResNetBackbone(name, native=True)
    assert list(ref.state_dict()) == list(hip.state_dict())
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():  # non-trivial BN parameters / running stats
        for p in ref.parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in ref.named_buffers():
            if n.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda()
    ref.train(train), hip.train(train)
    x = torch.rand(2, 3, size, size, generator=g)
    cots = None

    def run(m, dev):
        nonlocal cots
        xi = x.to(dev).requires_grad_(True)
        outs = m(xi)
        if cots is None:
            cots = [torch.randn(o.shape, generator=g) for o in outs[1:]]
        loss = sum((o * c.to(dev)).sum() for o, c in zip(outs[1:], cots))
        params = [p for _, p in m.named_parameters()]
        grads = torch.autograd.grad(loss, [xi] + params)
        return [o.detach().cpu() for o in outs], [t.detach().cpu() for t in grads]

    ro, rg = run(ref, "cpu")
    ho, hg = run(hip, "cuda")
    assert [tuple(o.shape) for o in ho] == [tuple(o.shape) for o in ro]
    # eval: 1e-4.  train: the deepest levels normalise with batch statistics over only 2*(size/32)^2 samples per
    # channel (8-32 here), which amplifies fp32 summation-order differences of 50 stacked layers: 1e-3 of scale.
    tol = 1e-3 if train else 1e-4
    for lvl, (a, b) in enumerate(zip(ho, ro)):
        torch.testing.assert_close(a, b, rtol=tol, atol=tol * max(1.0, float(b.abs().max())), msg=lambda s: f"level {lvl}: {s}")
    names = ["input"] + [n for n, _ in ref.named_parameters()]
    worst = {}
    for n, a, b in zip(names, hg, rg):
        # deep ReLU stacks: a few masks flip at fp32 rounding (see test_gpu_fullsize.py); rms criterion.
        # The stem (conv1 / bn1 / input) sits below torch's max-pool, whose backward routes the gradient of tied
        # maxima (post-ReLU zeros) to different taps on CPU and GPU - both valid - so it gets the loose bound.
        err = (a - b).abs()
        rms = float(err.pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-12))
        stem = n == "input" or n.startswith("model.conv1") or n.startswith("model.bn1")
        worst["stem" if stem else "stages"] = max(worst.get("stem" if stem else "stages", 0.0), rms)
        # Measured three ways on resnet50 @128 (tools/bb_debug.py): PyTorch-GPU(MIOpen) vs PyTorch-CPU differ
        # by 2.1-2.5e-3 rms on early-layer gradients, the HIP path by 3.2-4.5e-3 from CPU and 2.5-3.4e-3 from
        # PyTorch-GPU, while the last block agrees to 1e-6 everywhere: one flipped ReLU mask among layer4's 65k
        # elements moves every upstream gradient by ~1/sqrt(65k) = 4e-3.  Bound = that noise floor with margin.
        assert rms < (5e-2 if (train or stem) else 2e-2), f"{n}: rms-rel {rms:.2e}"
    print(f"{name} size {size} train={train}: worst gradient rms-rel {worst}")
    if train:
        hs, rs = hip.state_dict(), ref.state_dict()
        for k in rs:
            if "running_" in k:
                torch.testing.assert_close(hs[k].cpu(), rs[k], rtol=1e-3, atol=1e-4, msg=lambda s: f"{k}: {s}")


def test_native_backbone_bf16_close():
    import sihl_amd

    torch.manual_seed(0)
    m = sihl_amd.ResNetBackbone("resnet50", native=True).cuda().eval()
    x = torch.rand(2, 3, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m(x)  # fp32 native
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x)
    for lvl in range(1, 6):
        a, b = out[lvl].float(), ref[lvl]
        assert float((a - b).abs().max() / b.abs().max()) < 6e-2, lvl
