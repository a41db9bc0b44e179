"""The caller of the hot path against a fixture produced by the reference's OWN
``SihlLightningModule.configure_optimizers`` / ``training_step`` (tests/golden/make_golden_caller.py runs
src/sihl/lightning_module.py:68-120,179-245 behind a `lightning` stand-in): parameter groups, LinearLR warm-up +
scheduler sequence, the summed loss, every parameter's gradient norm and the parameters after AdamW steps."""
import numpy as np
import pytest
import torch

import oracle
from util import CALLER, TinyBackbone, caller_batch, golden_state_dict, load_npz

from sihl_amd.train import Trainer


def _build(ns, device="cpu"):
    bb = TinyBackbone()
    neck = ns.BiFPN(bb.out_channels, CALLER["neck_channels"], CALLER["bottom"], CALLER["top"], num_layers=1)
    head = ns.ObjectDetection(neck.out_channels, CALLER["num_classes"], CALLER["bottom"], CALLER["top"],
                              num_channels=CALLER["neck_channels"])
    return bb, neck, head


def _trainer(model, **kw):
    return Trainer(model, grad_clip_norm=None, scheduler=torch.optim.lr_scheduler.CosineAnnealingLR,
                   scheduler_kwargs={"T_max": CALLER["t_max"], "warmup": CALLER["warmup"]}, **CALLER["opt"], **kw)


def _check_groups(data, model, trainer):
    names = list(data["group.names"])
    assert sorted(n for n, _ in model.named_parameters()) == names  # same parameter names as the reference model
    where = {}
    for g in trainer.optimizer.param_groups:
        for p in g["params"]:
            where[id(p)] = g
    params = dict(model.named_parameters())
    for n, lr, wd in zip(names, data["group.base_lr"], data["group.weight_decay"]):
        g = where[id(params[n])]
        assert g.get("initial_lr", g["lr"]) == pytest.approx(float(lr), rel=1e-12), n
        assert g["weight_decay"] == pytest.approx(float(wd), rel=1e-12), n
    return names, params, where


def test_trainer_reproduces_reference_caller_on_cpu():
    data = load_npz("caller_step")
    bb, neck, head = _build(oracle)
    model = oracle.SihlModel(bb, neck, [head])
    model.load_state_dict(golden_state_dict(data), strict=True)
    trainer = _trainer(model)
    names, params, where = _check_groups(data, model, trainer)
    # one training step's loss and gradients (lightning_module.py:88-108: one extract_features, sum of head losses)
    model.train()
    x, target = caller_batch()
    loss, _ = trainer.forward_loss(x, [target])
    loss.backward()
    assert float(loss) == pytest.approx(float(data["step.loss"]), rel=1e-5)
    got = np.array([float(params[n].grad.norm()) for n in names])
    np.testing.assert_allclose(got, data["step.grad_norm"], rtol=2e-4, atol=2e-5)  # atol: the conv bias in front of a BatchNorm has a zero gradient up to rounding (1e-5)
    # LinearLR(0.01 -> 1, warmup steps) then the scheduler, stepped once per optimizer step (:226-241)
    for s in range(CALLER["sched_steps"]):
        lrs = np.array([where[id(params[n])]["lr"] for n in names])
        np.testing.assert_allclose(lrs, data["sched.lr_by_param"][s], rtol=1e-10)
        trainer.optimizer.step()
        trainer._step_scheduler()
    sums = np.array([float(params[n].detach().double().sum()) for n in names])
    # (Adam normalises the gradient: a parameter whose true gradient is zero - the conv bias in front of a BatchNorm -
    # moves by +-lr per step in the direction of its rounding noise, so it is left out)
    real = data["step.grad_norm"] > 1e-4
    assert real.sum() >= len(names) - 4
    np.testing.assert_allclose(sums[real], data["step.param_sum_after"][real], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_hip_path_reproduces_reference_caller():
    """Same fixture, neck and head on the HIP kernels (fp32, the parity configuration), backbone stand-in on
    PyTorch-ROCm: loss within 1e-4, gradient norms within 1e-3 of the reference's own numbers."""
    import types

    import sihl_amd

    data = load_npz("caller_step")
    ns = types.SimpleNamespace(BiFPN=sihl_amd.layers.BiFPN, ObjectDetection=sihl_amd.heads.ObjectDetection)
    bb, neck, head = _build(ns)
    model = sihl_amd.SihlModel(bb, neck, [head])
    model.load_state_dict(golden_state_dict(data), strict=True)
    model = model.cuda()
    trainer = _trainer(model, wgrad_stream="off")
    names, params, where = _check_groups(data, model, trainer)
    model.train()
    x, target = caller_batch()
    target = {k: [t.cuda() for t in v] for k, v in target.items()}
    loss, _ = trainer.forward_loss(x.cuda(), [target])
    loss.backward()
    assert float(loss) == pytest.approx(float(data["step.loss"]), rel=1e-4)
    got = np.array([float(params[n].grad.norm()) for n in names])
    np.testing.assert_allclose(got, data["step.grad_norm"], rtol=1e-3, atol=5e-5)
