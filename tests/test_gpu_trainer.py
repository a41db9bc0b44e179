"""GPU: the HIP-graph training step (Trainer(graph=True)) against the same steps launched eagerly."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    import sihl_amd
    torch.manual_seed(0)
    backbone = sihl_amd.ResNetBackbone("resnet18", top_level=5)
    neck = sihl_amd.layers.BiFPN(backbone.out_channels, 32, 3, 6, num_layers=1)
    head = sihl_amd.heads.ObjectDetection(neck.out_channels, num_classes=5, bottom_level=3, top_level=6, num_channels=32)
    return sihl_amd.SihlModel(backbone, neck, [head]).cuda().to(memory_format=torch.channels_last)


def _batch(seed, n_boxes):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(len(n_boxes), 3, 128, 128, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    boxes, classes = [], []
    for n in n_boxes:
        xy = torch.rand(n, 2, generator=g) * 70
        wh = 20 + torch.rand(n, 2, generator=g) * 30
        boxes.append(torch.cat([xy, xy + wh], 1).cuda())
        classes.append(torch.randint(0, 5, (n,), generator=g).cuda())
    return images, [{"classes": classes, "boxes": boxes}]


# Training is chaotic: a last-bit difference (atomic accumulation order in index_add's backward) is amplified by Adam's
# normalisation into a different lr-sized step for noise-dominated elements.  A small lr keeps the two trajectories
# comparable over the handful of steps this test runs.
LR = 1e-5


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_graph_step_matches_eager(amp):
    from sihl_amd.train import Trainer
    ref_model = _model()
    graph_model = copy.deepcopy(ref_model)
    eager = Trainer(ref_model, lr=LR, grad_clip_norm=0.1, autocast_dtype=amp)
    graphed = Trainer(graph_model, lr=LR, grad_clip_norm=0.1, autocast_dtype=amp, graph=True)
    assert graphed.use_graph
    # same signature (3 images with 2, 0, 3 boxes) for every step, different contents per step
    for step in range(Trainer.GRAPH_WARMUP + 4):
        images, targets = _batch(step, (2, 0, 3))
        le, _ = eager.step(images, targets)
        lg, _ = graphed.step(images, targets)
        tol = 1e-4 if amp is None else 3e-2
        torch.testing.assert_close(lg.float(), le.float(), rtol=tol, atol=tol, msg=lambda s: f"step {step}: {s}")
    assert len(graphed._graphs) == 1  # the later steps replayed one captured graph
    # every element within a few lr-sized steps, and the tensors equal on average to far below one step
    max_tol, mean_tol = (3 * LR, 0.1 * LR) if amp is None else (2e-3, 2e-4)
    for (n, a), b in zip(ref_model.named_parameters(), graph_model.parameters()):
        d = (a.detach().float() - b.detach().float()).abs()
        assert float(d.max()) <= max_tol, (n, float(d.max()))
        assert float(d.mean()) <= mean_tol, (n, float(d.mean()))
    tol = 1e-3 if amp is None else 2e-2
    for (n, a), b in zip(ref_model.named_buffers(), graph_model.buffers()):
        torch.testing.assert_close(b.float(), a.float(), rtol=tol, atol=tol, msg=lambda s: f"{n}: {s}")


def test_graph_trainer_new_signature_runs_eagerly_then_captures():
    from sihl_amd.train import Trainer
    tr = Trainer(_model(), lr=1e-3, graph=True)
    for n_boxes in ((1, 2), (1, 2), (1, 2), (3, 0), (1, 2)):
        loss, _ = tr.step(*_batch(0, n_boxes))
        assert torch.isfinite(loss)
    assert len(tr._graphs) == 1 and tr._seen[next(iter(tr._graphs))] >= Trainer.GRAPH_WARMUP


def test_graph_step_follows_lr_schedule():
    """In graph mode the learning rates live in device tensors the schedule rewrites before each replay: same
    values as the eager scheduler, and the replayed AdamW really uses them (lr 0 during the first steps freezes the
    weights; a later non-zero lr moves them)."""
    from sihl_amd.train import Trainer
    sched = torch.optim.lr_scheduler.LambdaLR
    kw = {"lr_lambda": lambda step: 0.0 if step < 4 else 1.0}
    ref_model = _model()
    graph_model = copy.deepcopy(ref_model)
    eager = Trainer(ref_model, lr=LR, grad_clip_norm=0.1, scheduler=sched, scheduler_kwargs=dict(kw))
    graphed = Trainer(graph_model, lr=LR, grad_clip_norm=0.1, graph=True, scheduler=sched, scheduler_kwargs=dict(kw))
    w0 = graph_model.neck.layers[0].up_convs[0][0].weight.detach().clone()
    for step in range(7):
        images, targets = _batch(step, (2, 0, 3))
        eager.step(images, targets)
        graphed.step(images, targets)
        for ge, gg in zip(eager.optimizer.param_groups, graphed.optimizer.param_groups):
            assert abs(float(gg["lr"]) - float(ge["lr"])) <= 1e-12 + 1e-6 * float(ge["lr"]), (step, ge["lr"], gg["lr"])
        moved = not torch.equal(graph_model.neck.layers[0].up_convs[0][0].weight, w0)
        # steps 0..3 run with lr 0 (steps 2.. are graph replays): weights must not move until the schedule says so
        assert moved == (step >= 4), (step, moved)
    for a, b in zip(ref_model.parameters(), graph_model.parameters()):
        assert float((a - b).abs().max()) <= 3 * LR


def test_object_detection_validation_reports_coco_map():
    """validation_step / on_validation_end of the HIP head (reference :219-250): loss + COCO-protocol box mAP keys."""
    model = _model().eval()
    head = model.heads[0]
    head.on_validation_start()
    with torch.no_grad():
        for seed in range(2):
            images, targets = _batch(seed, (2, 0, 3))
            loss, _ = head.validation_step(model.extract_features(images), **targets[0])
            assert torch.isfinite(loss)
    metrics = head.on_validation_end()
    for key in ("map", "map_50", "map_75", "mar_1", "mar_10", "mar_100", "loss"):
        assert key in metrics
    assert 0.0 <= metrics["map"] <= 1.0 and metrics["loss"] > 0


@pytest.mark.parametrize("mode", ["small", "all"])
def test_wgrad_side_stream_gives_identical_gradients(mode):
    """Weight gradients launched on the second HIP stream (ops.wgrad_side_stream) are the same kernels on the same
    operands: every gradient reproducible run to run and equal to the single-stream backward up to the fp32 summation
    order of the K-splits, repeatedly (a missing fork / join edge or an operand freed too early would show as a
    mismatch under the allocator's block reuse)."""
    from sihl_amd import ops
    from sihl_amd.train import Trainer
    model = _model()
    tr = Trainer(model, lr=LR, wgrad_stream="off")
    images, targets = _batch(3, (2, 1, 3))
    model.train()

    def grads(m):
        tr.optimizer.zero_grad(set_to_none=True)
        tr.wgrad_stream = m
        loss, _ = tr.forward_loss(images, targets)
        tr._backward(loss)
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    ref, ref2 = grads("off"), grads("off")
    exact = all(torch.equal(ref[n], ref2[n]) for n in ref)  # (atomics in the loss backward may reorder last bits)
    first = grads(mode)
    assert set(first) == set(ref)
    for _ in range(3):
        got = grads(mode)
        for n in ref:
            # beside the dgrad chain the K-splits are fewer and longer (other fp32 summation order): the same run twice
            # is bit-identical, against the single-stream gradients it agrees to fp32 rounding
            if exact:
                assert torch.equal(got[n], first[n]), n
            scale = float(ref[n].abs().max()) + 1e-12
            assert float((got[n] - ref[n]).abs().max()) <= 1e-4 * scale, n
    assert ops._SIDE is None  # the context manager restored the single-stream state


def test_side_stream_with_a_module_used_twice_per_step():
    """A parameter used twice in one step: autograd ADDS its two gradients (a main-stream kernel) while the side stream may
    still be writing them - ops._guard_shared_parameters makes the main stream wait from the second use on.  The gradients
    of a conv block and an MLP applied to two inputs must equal the single-stream ones bit for bit, run after run."""
    import sihl_amd
    from sihl_amd import ops
    torch.manual_seed(4)
    block = sihl_amd.layers.ConvNormAct(64, 64, 3).cuda().to(memory_format=torch.channels_last)
    mlp = sihl_amd.heads.MLP(64, [64, 64, 8], norm_layer=torch.nn.LayerNorm, activation_layer=torch.nn.SiLU).cuda()
    xs = [torch.randn(8, 64, 96, 96, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(2)]
    params = list(block.parameters()) + list(mlp.parameters())

    def grads(mode):
        for p in params:
            p.grad = None
        ya, yb = block(xs[0]), block(xs[1])          # the same weights twice
        za = mlp(ya.permute(0, 2, 3, 1).reshape(-1, 64)[:4096])
        zb = mlp(yb.permute(0, 2, 3, 1).reshape(-1, 64)[:4096])
        loss = (za.float() ** 2).mean() + (zb.float() ** 2).mean() + ya.float().mean() + yb.float().abs().mean()
        with ops.wgrad_side_stream(mode):
            loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    ref, first = grads("off"), grads("all")
    for _ in range(5):
        for a, f, b in zip(grads("all"), first, ref):
            assert torch.equal(a, f)  # reproducible run to run (a race with the side stream would not be) ...
            # ... and equal to the single-stream gradients up to the fp32 summation order of the K-splits
            assert float((a - b).abs().max()) <= 1e-4 * (float(b.abs().max()) + 1e-12)


def test_side_stream_is_refused_for_weights_autograd_would_copy():
    """Weight gradients written by the side stream are handed to autograd while that stream may still be writing them,
    which is safe only while AccumulateGrad takes the tensor as it is.  A model whose conv weights are NOT stored
    channels_last (model.cuda() without .to(memory_format=channels_last)) would make autograd launch a layout copy on the
    main stream: the Trainer must keep such a model single-stream (with a warning) - and its gradients must equal the
    channels_last model's."""
    import warnings

    import sihl_amd
    from sihl_amd.train import Trainer

    ref_model = _model()  # channels_last
    torch.manual_seed(0)
    backbone = sihl_amd.ResNetBackbone("resnet18", top_level=5)
    neck = sihl_amd.layers.BiFPN(backbone.out_channels, 32, 3, 6, num_layers=1)
    head = sihl_amd.heads.ObjectDetection(neck.out_channels, num_classes=5, bottom_level=3, top_level=6, num_channels=32)
    nchw_model = sihl_amd.SihlModel(backbone, neck, [head]).cuda()  # weights left NCHW-contiguous
    nchw_model.load_state_dict(ref_model.state_dict())
    assert any(p.dim() == 4 and p.shape[2] > 1 and not p.is_contiguous(memory_format=torch.channels_last)
               for p in nchw_model.parameters())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr = Trainer(nchw_model, lr=LR, wgrad_stream="all")
    assert tr.wgrad_stream == "off" and any("channels_last" in str(x.message) for x in w)
    ref_tr = Trainer(ref_model, lr=LR, wgrad_stream="all")
    assert ref_tr.wgrad_stream == "all"
    images, targets = _batch(3, (2, 1, 3))
    out = []
    for t, m in ((tr, nchw_model), (ref_tr, ref_model)):
        m.train()
        t.optimizer.zero_grad(set_to_none=True)
        loss, _ = t.forward_loss(images, targets)
        t._backward(loss)
        torch.cuda.synchronize()
        out.append({n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert set(out[0]) == set(out[1])
    for n in out[0]:
        scale = float(out[1][n].abs().max()) + 1e-12
        assert float((out[0][n] - out[1][n]).abs().max()) <= 1e-4 * scale, n


def test_multi_gpu_code_path_on_one_rank_matches_plain_step():
    """The data-parallel step (gradient buckets filled from grad-ready hooks, packed on the wgrad side stream, RCCL
    all-reduce on the collective's stream, averages scattered back) run in a process group of ONE rank must train
    exactly like the plain step: averaging over one replica is the identity."""
    import torch.distributed as dist
    from sihl_amd.train import Trainer
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        ref_model = _model()
        dp_model = copy.deepcopy(ref_model)
        plain = Trainer(ref_model, lr=LR, grad_clip_norm=0.1, wgrad_stream="off")
        dp = Trainer(dp_model, lr=LR, grad_clip_norm=0.1, wgrad_stream="all", force_buckets=True, bucket_mb=4.0)
        assert dp.averager.active and len(dp.averager.buckets) > 1
        for step in range(4):
            images, targets = _batch(step, (2, 0, 3))
            lp, _ = plain.step(images, targets)
            ld, _ = dp.step(images, targets)
            torch.testing.assert_close(ld.float(), lp.float(), rtol=1e-4, atol=1e-4)
        for (n, a), b in zip(ref_model.named_parameters(), dp_model.parameters()):
            d = (a.detach().float() - b.detach().float()).abs()
            assert float(d.max()) <= 3 * LR, (n, float(d.max()))
            assert float(d.mean()) <= 0.1 * LR, (n, float(d.mean()))
    finally:
        if created:
            dist.destroy_process_group()


def test_two_stream_step_reaches_allocator_steady_state():
    """With the host running ahead of the device (no synchronisation between steps) the two-stream eager step must not
    keep growing its memory: after a few steps the caching allocator stops calling hipMalloc.  (Holding the side stream's
    operands until an event behind its last kernel had COMPLETED kept a whole step's tensors alive into the next
    forward: reserved memory doubled and tripled in bursts of hipMalloc calls, each one a host stall.)"""
    from sihl_amd.train import Trainer

    model = _model()
    tr = Trainer(model, lr=LR, autocast_dtype=torch.bfloat16, wgrad_stream="all")
    images, targets = _batch(3, [3, 0, 5, 2, 4, 1, 2, 3])
    for _ in range(6):
        tr.step(images, targets)
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    for _ in range(12):  # queued back to back: the host is several steps ahead of the device
        tr.step(images, targets)
    torch.cuda.synchronize()
    st1 = torch.cuda.memory_stats()
    assert st1["num_device_alloc"] - st0["num_device_alloc"] <= 2, (st0["num_device_alloc"], st1["num_device_alloc"])
    assert st1["reserved_bytes.all.current"] <= st0["reserved_bytes.all.current"] * 1.05 + (64 << 20)


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_eval_after_graph_replays_sees_the_trained_weights(amp):
    """Round-3 advisor finding: a HIP-graph replay rewrites weights, gamma / beta and running statistics by raw pointer, so
    the inference-side caches (eval BatchNorm affines, operand casts, MLP plans - keyed on tensor versions) went stale in a
    train -> validate -> train -> validate loop.  Trainer.step now bumps a parameter generation the cache keys hold: eval
    after replays must equal the eval of an eagerly trained twin, both times."""
    from sihl_amd.train import Trainer
    ref_model = _model()
    graph_model = copy.deepcopy(ref_model)
    # (lr: large enough that a validation visibly differs from the previous one, small enough that the eager and the replayed
    # trajectory - Adam turns last-bit gradient differences into lr-sized steps on noise-dominated weights - stay comparable)
    eager = Trainer(ref_model, lr=2e-4, grad_clip_norm=0.1, autocast_dtype=amp)
    graphed = Trainer(graph_model, lr=2e-4, grad_clip_norm=0.1, autocast_dtype=amp, graph=True)
    probe, _ = _batch(99, (1, 1, 1))

    def evaluate(model):
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=amp or torch.float32, enabled=amp is not None):
            levels = model.backbone(probe)
            if amp is not None:
                levels = [t if i == 0 else t.to(amp) for i, t in enumerate(levels)]
            feats = model.neck(levels)
            head = model.heads[0]
            flat = head._flat_features(feats) if hasattr(head, "_flat_features") else None
            out = head(feats)
        model.train()
        return [feats[-1].float().clone(), out[1].float().clone()] + ([] if flat is None else [flat.float().clone()])

    step = 0
    prev = None
    for phase in range(2):
        for _ in range(Trainer.GRAPH_WARMUP + 2 if phase == 0 else 3):
            images, targets = _batch(step, (2, 0, 3))
            eager.step(images, targets)
            graphed.step(images, targets)
            step += 1
        assert len(graphed._graphs) == 1
        a, b = evaluate(ref_model), evaluate(graph_model)
        tol = 5e-3 if amp is None else 5e-2
        for x, y in zip(a, b):
            torch.testing.assert_close(y, x, rtol=tol, atol=tol, msg=lambda s: f"validation {phase}: {s}")
        if prev is not None:  # the second validation is NOT the first one again
            assert float((a[1] - prev[1]).abs().max()) > 0 and float((b[1] - prev[1]).abs().max()) > 0
        prev = b


def test_sibling_heads_report_their_validation_metrics():
    """InstanceSegmentation / KeypointDetection validation (reference instance_segmentation.py:299-330,
    keypoint_detection.py:320-350): loss + COCO-protocol mask mAP / PCK@0.05, accumulated over two steps; the metric
    arithmetic itself is hand-checked on the CPU (tests/test_host_cpu.py)."""
    import sihl_amd
    chans = [3, 8, 16, 32, 32, 32]
    g = torch.Generator().manual_seed(2)
    feats = [torch.rand(2, 3, 128, 128, generator=g).cuda()] + [torch.randn(2, c, 128 // 2 ** l, 128 // 2 ** l, generator=g).cuda()
                                                                 for l, c in enumerate(chans) if l > 0]
    torch.manual_seed(0)
    iseg = sihl_amd.heads.InstanceSegmentation(chans, num_classes=3, num_channels=32, max_instances=20).cuda().eval()
    masks = [torch.zeros(2, 128, 128, dtype=torch.bool), torch.zeros(1, 128, 128, dtype=torch.bool)]
    masks[0][0, 10:60, 20:70] = True
    masks[0][1, 70:120, 50:110] = True
    masks[1][0, 30:100, 30:100] = True
    classes = [torch.tensor([0, 2]), torch.tensor([1])]
    iseg.on_validation_start()
    for _ in range(2):
        loss, _ = iseg.validation_step(feats, [c.cuda() for c in classes], [m.cuda() for m in masks])
        assert torch.isfinite(loss)
    out = iseg.on_validation_end()
    assert {"map", "map_50", "map_75", "mar_1", "mar_10", "mar_20", "loss"} <= set(out) and -1.0 <= out["map"] <= 1.0

    kpt = sihl_amd.heads.KeypointDetection(chans, num_keypoints=4, num_channels=32, max_instances=10).cuda().eval()  # 16 positions at level 5
    keypoints = [torch.tensor([[[20., 20], [40, 30], [60, 50], [30, 70]], [[80., 90], [100, 100], [90, 110], [70, 120]]]),
                 torch.tensor([[[64., 64], [70, 60], [50, 80], [90, 40]]])]
    presence = [torch.tensor([[True, True, True, False], [True, True, True, True]]), torch.tensor([[True, False, True, True]])]
    kpt.on_validation_start()
    for _ in range(2):
        loss, _ = kpt.validation_step(feats, [k.cuda() for k in keypoints], [p.cuda() for p in presence])
        assert torch.isfinite(loss)
    out = kpt.on_validation_end()
    assert set(out) == {"PCK", "loss"} and 0.0 <= out["PCK"] <= 1.0
    assert kpt.pck_computer.total == 2 * 10  # every visible ground-truth keypoint is counted once per step, matched or not



def test_graph_trainer_refuses_packet_capture_replays():
    """ROCm's graph packet capture lets other device allocations overwrite a replay's kernel arguments (sihl_amd/__init__.py;
    the graph-replay faults of rounds 1-4).  The package turns it off when it is imported before the first device call; a
    process that touched the GPU first cannot be fixed from inside, and Trainer(graph=True) must say so instead of replaying."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "SIHL_GRAPH_ENV_EARLY")}
    code = ("import torch; torch.zeros(8, device='cuda').add_(1); torch.cuda.synchronize(); import sihl_amd\n"
            "from sihl_amd.train import Trainer\n"
            "print('safe', sihl_amd.graph_replay_safe())\n"
            "m = sihl_amd.SihlModel(sihl_amd.ResNetBackbone('resnet18'), None, [sihl_amd.heads.SemanticSegmentation([3, 64, 64, 128, 256, 512], num_classes=3)]).cuda()\n"
            "try:\n    Trainer(m, graph=True)\n    print('constructed')\nexcept RuntimeError as e:\n    print('refused', 'packet capture' in str(e))\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=300)
    assert "safe False" in p.stdout and "refused True" in p.stdout, (p.stdout[-800:], p.stderr[-800:])
    code2 = "import sihl_amd, torch; torch.zeros(8, device='cuda'); print('safe', sihl_amd.graph_replay_safe())"
    p = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, cwd=root, env=env, timeout=300)
    assert "safe True" in p.stdout, (p.stdout[-800:], p.stderr[-800:])


def test_full_size_graph_replays_survive_other_allocations():
    """The north-star training step (ResNet50 + BiFPN + ObjectDetection, bs 32, 512^2, bf16) as ONE HIP graph, replayed five
    times with ATen work between the replays (the L2 norm of all parameters) and 72 fresh NaN-filled allocations after the
    fourth step: the losses must be the eager run's.  On ROCm's graph packet-capture path this exact sequence computed inf
    (or faulted the GPU: the graph-replay fault of rounds 1-4, profiles/r04_graph_replay_root_cause.txt); the package default
    turns that path off.  Runs tools/graph_bisect.py in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "graph_bisect.py"), "default", "norms", "checksums"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert "[default] OK" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    losses = [float(l.split("loss ")[1].split(",")[0]) for l in p.stdout.splitlines() if "] step " in l]
    want = [61.1518, 53.4959, 42.4053, 38.2965, 33.6236, 30.8844, 28.4207]  # the eager two-stream / single-stream trajectory
    assert len(losses) == 7 and all(abs(a - b) < 0.5 for a, b in zip(losses, want)), losses
    assert "changed by 72 fresh NaN-filled eager allocations: []" in p.stdout


def test_derived_weights_keep_their_gradient_on_the_main_stream():
    """A conv whose weight reaches the kernel through autograd ops - the zero-padded copy of an odd channel count
    (layers/convblocks.py: 21 segmentation classes -> 24) - must NOT run its weight gradient on the side stream: autograd slices
    that gradient on the main stream at once, before the join.  Round 4 found the race as garbage gradient norms in the
    two-stream SemanticSegmentation configurations (profiles/r04_side_stream_leaf_race.txt).  Leaf weights still go to the side
    stream; and a two-stream run of a 21-class segmentation model reproduces the single-stream trajectory."""
    import sihl_amd
    from sihl_amd import ops
    from sihl_amd.layers.convblocks import ConvNormAct
    from sihl_amd.train import Trainer
    torch.manual_seed(0)
    x = torch.randn(2, 16, 32, 32, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for cout, expect_side in ((21, False), (24, True)):
        blk = ConvNormAct(16, cout, kernel_size=1, norm=None, act=None).cuda().to(memory_format=torch.channels_last)
        with ops.wgrad_side_stream("all", x.device):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x)
            y.float().square().mean().backward()
            used = ops.side_stream_in_use() is not None
        assert used == expect_side, (cout, used)
        assert torch.isfinite(blk[0].weight.grad).all() and torch.isfinite(blk[0].bias.grad).all()

    def make():
        torch.manual_seed(1)
        bb = sihl_amd.ResNetBackbone("resnet18", top_level=5)
        neck = sihl_amd.layers.FPN(bb.out_channels, 64, 3, 5)
        head = sihl_amd.heads.SemanticSegmentation(neck.out_channels, num_classes=21, bottom_level=3, top_level=5, num_channels=64)
        return sihl_amd.SihlModel(bb, neck, [head]).cuda().to(memory_format=torch.channels_last)

    g = torch.Generator().manual_seed(2)
    images = torch.rand(4, 3, 128, 128, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    target = torch.randint(0, 21, (4, 128, 128), generator=g).cuda()
    runs = {}
    for mode in ("all", "off", "all"):
        tr = Trainer(make(), lr=1e-3, grad_clip_norm=0.1, autocast_dtype=torch.bfloat16, wgrad_stream=mode)
        runs.setdefault(mode, []).append([float(tr.step(images, [target])[0]) for _ in range(6)])
    assert runs["all"][0] == runs["all"][1], runs["all"]  # the same bits twice
    for a, b in zip(runs["all"][0], runs["off"][0]):
        assert abs(a - b) < 2e-2 * max(1.0, abs(b)), (runs["all"][0], runs["off"][0])
