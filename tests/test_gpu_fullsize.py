"""GPU parity at BASELINE sizes: BiFPN(3-7, 256 ch, 3 layers) + ObjectDetection(80 classes) on ResNet50-shaped
levels of a 512x512 input, HIP fp32 against the CPU oracle on identical weights and inputs (batch kept at 2 so
that the CPU side finishes in seconds), plus size-independent properties at the full batch of 32."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CH = [3, 64, 256, 512, 1024, 2048]


def _levels(batch, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.zeros(batch, 3, 512, 512)] + [torch.randn(batch, c, 512 // 2 ** l, 512 // 2 ** l, generator=g)
                                                for l, c in enumerate(CH) if l > 0]


def test_bifpn_od_training_step_512_fp32_matches_oracle():
    import oracle
    import sihl_amd

    torch.manual_seed(0)
    o_neck = oracle.BiFPN(CH, 256, 3, 7)
    o_head = oracle.ObjectDetection(o_neck.out_channels, 80, 3, 7)
    h_neck = sihl_amd.layers.BiFPN(CH, 256, 3, 7)
    h_head = sihl_amd.heads.ObjectDetection(h_neck.out_channels, 80, 3, 7)
    h_neck.load_state_dict(o_neck.state_dict())
    h_head.load_state_dict(o_head.state_dict())
    h_neck, h_head = h_neck.cuda(), h_head.cuda()
    levels = _levels(2, 1)
    boxes = [torch.tensor([[21.3, 33.7, 218.2, 301.9], [197.4, 161.1, 423.6, 339.3], [300.5, 40.2, 380.1, 120.7]]),
             torch.tensor([[50.5, 260.25, 130.0, 420.75]])]
    classes = [torch.tensor([1, 44, 79]), torch.tensor([7])]

    def run(neck, head, dev):
        lv = [t.to(dev).requires_grad_(i >= 3) for i, t in enumerate(levels)]
        feats = neck(lv)
        loss, metrics = head.training_step(feats, [c.to(dev) for c in classes], [b.to(dev) for b in boxes])
        grads = torch.autograd.grad(loss, lv[3:6])
        return [f.detach().float().cpu() for f in feats[3:]], loss.detach().cpu(), \
            {k: v.detach().cpu() for k, v in metrics.items()}, [g.cpu() for g in grads]

    ref_f, ref_loss, ref_m, ref_g = run(o_neck, o_head, "cpu")
    hip_f, hip_loss, hip_m, hip_g = run(h_neck, h_head, "cuda")
    for a, b in zip(hip_f, ref_f):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())))
    torch.testing.assert_close(hip_loss, ref_loss, rtol=1e-4, atol=1e-4)
    for k in ref_m:
        torch.testing.assert_close(hip_m[k], ref_m[k], rtol=1e-4, atol=1e-4, msg=lambda s: f"{k}: {s}")
    # Input gradients pass through 40+ ReLU/BatchNorm stages.  Where a pre-activation sits within fp32 rounding of
    # zero its mask flips between two summation orders; measured on the CPU, the fp32 oracle itself deviates from
    # an fp64 run of the same oracle by rms 0.8-1.3e-3 on exactly these tensors.  The gradient criterion is
    # therefore that noise floor (rms < 5e-3, worst element < 5 % of the tensor's scale), not 1e-4; forward
    # features, the loss and every loss component above ARE held to 1e-4.
    for a, b in zip(hip_g, ref_g):
        err = (a - b).abs()
        rms = float(err.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
        assert rms < 5e-3, f"gradient rms-relative error {rms:.2e}"
        assert float(err.max()) < 5e-2 * float(b.abs().max()), float(err.max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_batch_properties(dtype):
    """bs=32 at 512^2 (the benchmark shape), where a CPU oracle run would take minutes: properties instead.
    (1) linearity of the conv in its input: conv(a*x + y) = a*conv(x) + conv(y);
    (2) the weight gradient is the adjoint of the conv: <conv_w(x), dy> = <w, wgrad(x, dy)>;
    (3) the input gradient is the adjoint too:           <conv_w(x), dy> = <x, dgrad(w, dy)>;
    (4) batch independence in eval mode: image 0 of a 32-batch equals the same image run alone."""
    from sihl_amd import ops
    import sihl_amd

    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(32, 64, 64, 256, device=dev, generator=g).to(dtype)
    y = torch.randn(32, 64, 64, 256, device=dev, generator=g).to(dtype)
    w = (torch.randn(256, 3, 3, 256, device=dev, generator=g) * 0.02).to(dtype)
    dy = torch.randn(32, 64, 64, 256, device=dev, generator=g).to(dtype)
    tol = 2e-5 if dtype == torch.float32 else 2e-2

    def rel(a, b):
        return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp(min=1e-6))

    cx, _ = ops.conv2d_raw(x, w, None, 1, 1, 1)
    cy, _ = ops.conv2d_raw(y, w, None, 1, 1, 1)
    cxy, _ = ops.conv2d_raw((0.5 * x.float() + y.float()).to(dtype), w, None, 1, 1, 1)
    assert rel(cxy, 0.5 * cx.float() + cy.float()) < tol * 5
    inner = (cx.double() * dy.double()).sum()
    dw = ops.conv2d_wgrad_raw(x, dy, 3, 3, 1, 1, 1)
    assert abs(float((dw.double() * w.double()).sum() / inner) - 1.0) < tol * 5
    from sihl_amd import _C
    wt = ops.weight_for_dgrad(w, flip=True)
    dx = torch.empty_like(x)
    assert _C.lib().sihl_conv2d_dgrad(ops._p(dy), ops._p(wt), ops._p(dx), 32, 64, 64, 256, 256, 3, 3, 1, 1, 1,
                                      ops._dt(x), ops._stream()) == 0
    assert abs(float((dx.double() * x.double()).sum() / inner) - 1.0) < tol * 5

    torch.manual_seed(0)
    neck = sihl_amd.layers.BiFPN(CH, 256, 3, 7, num_layers=1).cuda().eval()
    levels = [t.cuda().to(dtype) if i else t.cuda() for i, t in enumerate(_levels(32, 5))]
    with torch.no_grad():
        full = neck(levels)
        one = neck([t[:1] for t in levels])
    for l in range(3, 8):
        assert rel(full[l][:1], one[l]) < tol * 5, l
