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

    def run(neck, head, dev, dtype=torch.float32):
        lv = [t.to(dev, dtype).requires_grad_(i >= 3) for i, t in enumerate(levels)]
        feats = neck(lv)
        loss, metrics = head.training_step(feats, [c.to(dev) for c in classes], [b.to(dev, dtype) for b in boxes])
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
    # That floor is MEASURED here, not quoted (round-3 review, weak 1c): the same oracle in fp64 on the same weights and
    # inputs is the yardstick, floor = how far the fp32 oracle lies from it; the HIP fp32 gradients must lie within 3x that
    # floor (+ 1e-4) of the fp64 gradients - and the floor itself must stay in the range that justifies the 5e-3 cap.
    import copy
    o64_neck, o64_head = copy.deepcopy(o_neck).double(), copy.deepcopy(o_head).double()
    torch.set_default_dtype(torch.float64)  # the oracle's anchor grids and constants follow the default dtype
    try:
        _, _, _, ref64_g = run(o64_neck, o64_head, "cpu", torch.float64)
    finally:
        torch.set_default_dtype(torch.float32)
    for a, b, b64 in zip(hip_g, ref_g, ref64_g):
        b64 = b64.float()
        norm = float(b64.pow(2).mean().sqrt())
        floor = float((b - b64).pow(2).mean().sqrt()) / norm
        rms64 = float((a - b64).pow(2).mean().sqrt()) / norm
        assert floor < 5e-3, f"fp32 oracle vs fp64 oracle: {floor:.2e} (the 5e-3 cap below assumes a floor near 1e-3)"
        assert rms64 < 3 * floor + 1e-4, f"gradient rms error vs fp64 {rms64:.2e}, fp32-oracle floor {floor:.2e}"
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


@pytest.mark.parametrize("shape", [(32, 128, 128, 128, 128), (32, 64, 64, 256, 256), (32, 32, 32, 512, 512)])
def test_strided_dgrad_parity_classes_equal_dilated_read_at_resnet_sizes(shape):
    """The 3x3 / stride-2 input gradient computed as four parity classes (the shipped path) and through the zero-dilated
    read of the whole output grid are the same sum of products in a different order: equal to bf16 rounding at the
    full ResNet50 stage-entry sizes of a 512x512 batch of 32, and every output pixel is written (no parity skipped)."""
    from sihl_amd import _C, ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(11)
    dy = torch.randn(N, H // 2, W // 2, Cout, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    wt = ops.weight_for_dgrad(w, flip=True)
    lib = _C.lib()
    outs = []
    for classes in (1, 0):
        lib.sihl_conv2d_strided_classes_enable(classes)
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = lib.sihl_conv2d_dgrad(ops._p(dy), ops._p(wt), ops._p(dx), N, H, W, Cin, Cout, 3, 3, 2, 1, 1, ops._dt(dx),
                                   ops._stream())
        assert rc == 0
        outs.append(dx.float())
    lib.sihl_conv2d_strided_classes_enable(1)
    a, b = outs
    assert torch.isfinite(a).all()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-2 * scale
    assert float((a - b).abs().mean()) <= 2e-3 * scale


def test_compact_projection_gradient_equals_dilated_dgrad_plus_add():
    """ResNet50 layer2 entry at bs 32: conv1's dgrad with the projection's compact gradient added at the strided
    pixels (sihl_conv2d_dgrad_add) against the two separate input gradients summed by PyTorch."""
    from sihl_amd import _C, ops
    N, H, W, Cin, C1, Cds = 32, 128, 128, 256, 128, 512
    g = torch.Generator(device="cuda").manual_seed(12)
    dz1 = torch.randn(N, H, W, C1, device="cuda", generator=g).to(torch.bfloat16)            # grad of conv1's output
    dzd = torch.randn(N, H // 2, W // 2, Cds, device="cuda", generator=g).to(torch.bfloat16)  # grad of the projection's
    w1 = (torch.randn(C1, 1, 1, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    wd = (torch.randn(Cds, 1, 1, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    lib = _C.lib()
    wt1, wtd = ops.weight_for_dgrad(w1, flip=True), ops.weight_for_dgrad(wd, flip=True)
    # reference: two plain dgrads (the projection's through the zero-dilated read), summed in fp32
    d1 = torch.empty((N, H, W, Cin), device="cuda", dtype=torch.bfloat16)
    d2 = torch.empty_like(d1)
    assert lib.sihl_conv2d_dgrad(ops._p(dz1), ops._p(wt1), ops._p(d1), N, H, W, Cin, C1, 1, 1, 1, 0, 1, ops._dt(d1),
                                 ops._stream()) == 0
    assert lib.sihl_conv2d_dgrad(ops._p(dzd), ops._p(wtd), ops._p(d2), N, H, W, Cin, Cds, 1, 1, 2, 0, 1, ops._dt(d2),
                                 ops._stream()) == 0
    ref = d1.float() + d2.float()
    # shipped path: compact projection gradient, then conv1's dgrad adds it
    dxs = torch.empty((N, H // 2, W // 2, Cin), device="cuda", dtype=torch.bfloat16)
    assert lib.sihl_conv2d_dgrad(ops._p(dzd), ops._p(wtd), ops._p(dxs), N, H // 2, W // 2, Cin, Cds, 1, 1, 1, 0, 1,
                                 ops._dt(dxs), ops._stream()) == 0
    dx = torch.empty_like(d1)
    assert lib.sihl_conv2d_dgrad_add(ops._p(dz1), ops._p(wt1), ops._p(dx), ops._p(dxs), 2, N, H, W, Cin, C1, 1, 1, 1, 0, 1,
                                     ops._dt(dx), None, 0, ops._stream()) == 0
    scale = float(ref.abs().max())
    assert float((dx.float() - ref).abs().max()) <= 2e-2 * scale
    assert float((d2.float()[:, 1::2] ).abs().max()) == 0.0  # the projection contributes nothing off its pixels
