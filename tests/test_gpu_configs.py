"""BASELINE.json configurations that are not the benchmark line, as parity / property tests:
  configs[1]  ResNet50 + FPN(3-5) + SemanticSegmentation (93 classes), 3x512x512, fp32 (batch 8 for the oracle leg, the
              configuration's batch 16 as a HIP-only training step; end-to-end tolerance derived from the head's conditioning)
  configs[4]  convnext_base-shaped TimmBackbone ([3, 3, 128, 256, 512, 1024] with the reference's fake level 1,
              timm_backbone.py:143-152) + BiFPN(3-7) + {ObjectDetection, SemanticSegmentation}, 3x640x640:
              fp32 small-batch parity against the oracle, bf16 batch-16 properties (P = 8525 positions)."""
import pytest
import torch
import torch.nn.functional as F


def test_timm_backbone_level_contract():
    """CPU: the contract lines of the reference (timm_backbone.py:143-152,161-186) - no GPU, no kernels."""
    import oracle
    import sihl_amd

    torch.manual_seed(0)
    o = oracle.TimmBackbone("convnext_base", depths=(1, 1, 1, 1), top_level=6)
    h = sihl_amd.TimmBackbone("convnext_base", depths=(1, 1, 1, 1), top_level=6)
    assert o.out_channels == h.out_channels == [3, 3, 128, 256, 512, 1024, 1024]
    assert tuple(h.dummy_input.shape) == (1, 3, 128, 128)  # 2^(top_level + 1)
    h.load_state_dict(o.state_dict(), strict=True)          # same parameter tree
    x = torch.rand(2, 3, 128, 192)
    levels = o(x)
    assert [tuple(t.shape[1:]) for t in levels] == [(3, 128, 192), (3, 64, 96), (128, 32, 48), (256, 16, 24),
                                                    (512, 8, 12), (1024, 4, 6), (1024, 2, 3)]
    assert levels[0] is x
    assert torch.equal(levels[1], F.interpolate(x, size=(64, 96)))  # fake level 1: nearest-resized input
    with pytest.raises(AssertionError):
        o(torch.rand(1, 3, 100, 128))  # sizes must divide by 2^top_level (:174-176)
    with pytest.raises(ValueError):
        sihl_amd.TimmBackbone("not_a_model")


def _close(a, b, tol, name):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    torch.testing.assert_close(a, b, rtol=tol, atol=tol * max(1.0, float(b.abs().max())), msg=lambda s: f"{name}: {s}")


@pytest.mark.gpu
def test_config2_resnet50_fpn_semseg_512_fp32_matches_oracle():
    import oracle
    import sihl_amd

    torch.manual_seed(0)
    o_bb = oracle.ResNetBackbone("resnet50", top_level=5)
    o_neck = oracle.FPN(o_bb.out_channels, 256, 3, 5)
    NCLS = 93  # the example's class count (BASELINE configs[1]; examples/semantic_segmentation.py)
    o_head = oracle.SemanticSegmentation(o_neck.out_channels, num_classes=NCLS, bottom_level=3, top_level=5)
    o_model = oracle.SihlModel(o_bb, o_neck, [o_head])
    h_bb = sihl_amd.ResNetBackbone("resnet50", top_level=5)
    h_neck = sihl_amd.layers.FPN(h_bb.out_channels, 256, 3, 5)
    h_head = sihl_amd.heads.SemanticSegmentation(h_neck.out_channels, num_classes=NCLS, bottom_level=3, top_level=5)
    h_model = sihl_amd.SihlModel(h_bb, h_neck, [h_head])
    h_model.load_state_dict(o_model.state_dict(), strict=True)
    h_model = h_model.cuda().to(memory_format=torch.channels_last)
    # batch 8 (16 in the config): train-mode BatchNorm inside the head's pyramid pooling normalises the 1x1-pooled map
    # over the batch alone, and with 2 samples that division amplifies a 1e-4 input difference twentyfold
    g = torch.Generator().manual_seed(5)
    B = 8
    x = torch.rand(B, 3, 512, 512, generator=g)
    target = torch.randint(0, NCLS, (B, 512, 512), generator=g)
    o_model.train(), h_model.train()

    def run(model, dev, feats=None):
        with torch.no_grad():  # forward quantities only: no autograd graph of a ResNet50 at 512^2 on the CPU
            if feats is None:
                feats = model.extract_features(x.to(dev))
            feats = [f.to(dev) for f in feats]
            loss, _ = model.heads[0].training_step(feats, target.to(dev))
            logits = model.heads[0].get_logits(feats)
        return [f.detach() for f in feats], logits.detach(), loss.detach()

    ref_f, ref_logits, ref_loss = run(o_model, "cpu")
    hip_f, hip_logits, hip_loss = run(h_model, "cuda")
    # (a) trunk + neck.  Behind the 50 conv + train-mode BatchNorm layers of the trunk the two summation orders differ by
    # up to 1.2e-4 of the feature scale on single elements (1 of 2 M measured): the features get 2e-4
    for l in range(3, 6):
        _close(hip_f[l], ref_f[l], 2e-4, f"FPN level {l}")
    # (b) the head, on IDENTICAL inputs: the oracle head on the HIP path's own features.  (This head is ill-conditioned
    # at random initialisation - train-mode BatchNorm over the batch of 1x1-pooled maps: measured on the oracle alone,
    # 1e-4 of feature noise moves its logits by 3e-3 - so end-to-end logits are bounded through (a) and (b), not 1e-4.)
    _, head_ref_logits, head_ref_loss = run(o_model, "cpu", feats=[f.float().cpu() for f in hip_f])
    _close(hip_logits, head_ref_logits, 1e-4, "logits (same features)")
    _close(hip_loss, head_ref_loss, 1e-4, "loss (same features)")
    # (c) end to end.  The tolerance is DERIVED here, not chosen: the reference's own head (the oracle, on the CPU) is run
    # once more on its own features with uniform noise of relative size 1e-4 injected at the neck output; the movement of
    # its logits / loss per unit of feature noise is the head's conditioning kappa on THIS input, and the end-to-end
    # deviation of the HIP path must stay within kappa x the feature deviation measured in (a) (x3: structured rounding
    # differences of a conv stack are not uniform noise) plus the head's own 1e-4 of (b).
    noise = 1e-4
    gn = torch.Generator().manual_seed(9)
    noisy = [f if l < 3 else f + noise * max(1.0, float(f.abs().max())) * (2 * torch.rand(f.shape, generator=gn) - 1)
             for l, f in enumerate(ref_f)]
    _, noisy_logits, noisy_loss = run(o_model, "cpu", feats=noisy)
    rel = lambda a, b: float((a.float().cpu() - b.float().cpu()).abs().max()) / max(1.0, float(b.abs().max()))  # noqa: E731
    kappa_logits, kappa_loss = rel(noisy_logits, ref_logits) / noise, rel(noisy_loss, ref_loss) / noise
    feat_err = max(rel(hip_f[l], ref_f[l]) for l in range(3, 6))
    e_logits, e_loss = rel(hip_logits, ref_logits), rel(hip_loss, ref_loss)
    print(f"config 2: feature deviation {feat_err:.2e}; head conditioning (logits / loss) {kappa_logits:.1f} / {kappa_loss:.1f}; "
          f"end-to-end deviation {e_logits:.2e} / {e_loss:.2e}")
    assert feat_err <= 2e-4
    assert e_logits <= 3 * kappa_logits * feat_err + 1e-4, (e_logits, kappa_logits, feat_err)
    assert e_loss <= 3 * kappa_loss * feat_err + 1e-4, (e_loss, kappa_loss, feat_err)
    assert e_logits <= 1e-2 and e_loss <= 1e-2  # and a hard backstop whatever the conditioning
    # the configuration's own batch (16) through the HIP path alone: one training step with finite loss and gradients
    x16 = torch.rand(16, 3, 512, 512, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    t16 = torch.randint(0, NCLS, (16, 512, 512), generator=g).cuda()
    loss16, _ = h_model.heads[0].training_step(h_model.extract_features(x16), t16)
    loss16.backward()
    assert torch.isfinite(loss16) and 0.5 * float(torch.log(torch.tensor(float(NCLS)))) < float(loss16) < 3 * float(torch.log(torch.tensor(float(NCLS))))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in h_model.parameters() if p.requires_grad)
    h_model.zero_grad(set_to_none=True)
    # eval forward: (scores, classes) at input resolution; classes are integers and must agree wherever the two best
    # logits are not within rounding of each other
    # validation metrics of the head (reference :94-120): accumulated on the device over two steps, against a direct
    # computation from the head's own class maps
    h_model.eval()
    hh = h_model.heads[0]
    hh.on_validation_start()
    with torch.no_grad():
        feats_a, feats_b = h_model.extract_features(x[:2].cuda()), h_model.extract_features(x[2:4].cuda())
        hh.validation_step(feats_a, target[:2].cuda())
        hh.validation_step(feats_b, target[2:4].cuda())
        cls = torch.cat([hh.forward(feats_a)[1], hh.forward(feats_b)[1]]).cpu()
    val = hh.on_validation_end()
    tt = target[:4]
    assert abs(val["pixel_accuracy"] - float((cls == tt).float().mean())) < 1e-9
    ious = []
    for c in range(NCLS):
        inter, union = ((cls == c) & (tt == c)).sum().item(), ((cls == c) | (tt == c)).sum().item()
        if union:
            ious.append(inter / union)
    assert abs(val["mean_iou"] - sum(ious) / len(ious)) < 1e-9 and val["loss"] == val["loss"]
    h_model.train()
    h_model.load_state_dict(o_model.state_dict(), strict=True)  # identical running statistics for the eval pass
    o_model.eval(), h_model.eval()
    with torch.no_grad():
        rs, rc = o_model(x[:2])[0]
        hs, hc = h_model(x[:2].cuda())[0]
    assert tuple(hs.shape) == tuple(rs.shape) == (2, 512, 512) and hc.dtype == rc.dtype == torch.int64
    _close(hs, rs, 1e-4, "scores")
    assert float((hc.cpu() != rc).float().mean()) < 1e-4


def _config5(ns_bb, ns_layers, ns_heads, model_cls, depths):
    bb = ns_bb("convnext_base", depths=depths, top_level=5)
    neck = ns_layers.BiFPN(bb.out_channels, 256, 3, 7)
    od = ns_heads.ObjectDetection(neck.out_channels, num_classes=80, bottom_level=3, top_level=7)
    ss = ns_heads.SemanticSegmentation(neck.out_channels, num_classes=21, bottom_level=3, top_level=5)
    return model_cls(bb, neck, [od, ss])


def _targets5(batch, size, dev, seed):
    g = torch.Generator().manual_seed(seed)
    boxes, classes = [], []
    for b in range(batch):
        n = 0 if b == 1 else 1 + b % 3
        xy = torch.rand(n, 2, generator=g) * (size * 0.6)
        wh = 32 + torch.rand(n, 2, generator=g) * (size * 0.3)
        boxes.append(torch.cat([xy, xy + wh], 1).to(dev))
        classes.append(torch.randint(0, 80, (n,), generator=g).to(dev))
    seg = torch.randint(0, 21, (batch, size, size), generator=g).to(dev)
    return [{"classes": classes, "boxes": boxes}, seg]


@pytest.mark.gpu
def test_config5_multitask_step_640_fp32_matches_oracle():
    import oracle
    import sihl_amd

    torch.manual_seed(0)
    o_model = _config5(oracle.TimmBackbone, oracle, oracle, oracle.SihlModel, (1, 1, 2, 1))
    h_model = _config5(sihl_amd.TimmBackbone, sihl_amd.layers, sihl_amd.heads, sihl_amd.SihlModel, (1, 1, 2, 1))
    assert h_model.backbone.out_channels == [3, 3, 128, 256, 512, 1024]
    h_model.load_state_dict(o_model.state_dict(), strict=True)
    h_model = h_model.cuda()
    o_model.train(), h_model.train()
    x = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(9))
    x = torch.cat([x, x.flip(3)], 0)  # batch 2 (image 1 has no boxes)

    def run(model, dev):
        tg = _targets5(2, 640, dev, 3)
        feats = model.extract_features(x.to(dev))
        losses = []
        for head, t in zip(model.heads, tg):
            loss, _ = head.training_step(feats, **t) if isinstance(t, dict) else head.training_step(feats, t)
            losses.append(loss)
        total = torch.stack(losses).sum()
        grads = torch.autograd.grad(total, [model.backbone.model.stem[0].weight, model.neck.lateral_connections[0][0].weight])
        return [f.detach() for f in feats[3:]], [l.detach() for l in losses], [g_.detach() for g_ in grads]

    ref_f, ref_l, ref_g = run(o_model, "cpu")
    hip_f, hip_l, hip_g = run(h_model, "cuda")
    assert [tuple(f.shape[2:]) for f in hip_f] == [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]  # P = 8525
    for l, (a, b) in enumerate(zip(hip_f, ref_f)):
        _close(a, b, 1e-4, f"BiFPN level {l + 3}")
    for name, a, b in zip(("detection loss", "segmentation loss"), hip_l, ref_l):
        _close(a, b, 1e-4, name)
    for name, a, b in zip(("stem weight gradient", "lateral weight gradient"), hip_g, ref_g):
        err = float((a.cpu() - b).norm() / b.norm())
        assert err < 1e-2, (name, err)  # through ~40 ReLU / BatchNorm stages: the fp32 noise floor (test_gpu_fullsize)


@pytest.mark.gpu
def test_config5_multitask_bf16_batch16_properties():
    import sihl_amd
    from sihl_amd.train import Trainer

    torch.manual_seed(0)
    # the configuration's real trunk depth: convnext_base = (3, 3, 27, 3) blocks (timm_backbone.py:119-126)
    model = _config5(sihl_amd.TimmBackbone, sihl_amd.layers, sihl_amd.heads, sihl_amd.SihlModel, (3, 3, 27, 3))
    assert sum(p.numel() for p in model.backbone.parameters()) > 80e6  # ~88 M parameters: the base model, not a stand-in
    model = model.cuda().to(memory_format=torch.channels_last)
    x = torch.rand(16, 3, 640, 640, generator=torch.Generator().manual_seed(2)).cuda()
    x = x.contiguous(memory_format=torch.channels_last)
    targets = _targets5(16, 640, "cuda", 4)
    tr = Trainer(model, lr=1e-4, weight_decay=1e-4, backbone_lr_factor=0.1, grad_clip_norm=0.1,
                 autocast_dtype=torch.bfloat16)
    # the summed loss is the sum of the heads' losses on ONE extract_features pass (lightning_module.py:88-108)
    model.train()
    total, metrics = tr.forward_loss(x, targets)
    parts = [metrics["head0/train/location_loss"] + 10 * metrics["head0/train/box_loss"] +
             metrics["head0/train/class_loss"] + metrics["head0/train/iou_loss"]]
    assert torch.isfinite(total)
    assert float(total) > float(parts[0]) > 0  # detection part + a positive segmentation part
    total.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    n_grad = sum(p.grad is not None for p in model.parameters())
    assert n_grad == sum(p.requires_grad for p in model.parameters())  # both heads and the whole trunk got gradients
    losses = [float(tr.step(x, targets)[0]) for _ in range(4)]
    assert all(l == l and l < 1e4 for l in losses)
    assert losses[-1] < losses[0]  # four AdamW steps on one batch reduce its loss
    # eval: batch independence of both heads' outputs and the anchor count of the 640x640 pyramid
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        levels = model.backbone(x)
    levels = [t if i == 0 else t.to(torch.bfloat16) for i, t in enumerate(levels)]
    with torch.no_grad():
        feats = model.neck(levels)
        one = model.neck([t[:1] for t in levels])
        assert sum(f.shape[2] * f.shape[3] for f in feats[3:]) == 8525
        for l in range(3, 8):
            d = float((feats[l][:1].float() - one[l].float()).abs().max())
            assert d <= 2e-2 * float(one[l].float().abs().max()), l
        num, scores, classes, boxes = model.heads[0](feats)
        assert tuple(boxes.shape) == (16, 100, 4) and classes.dtype == torch.int64
        s, c = model.heads[1](feats)
        assert tuple(s.shape) == (16, 640, 640) and c.dtype == torch.int64 and int(c.max()) < 21
