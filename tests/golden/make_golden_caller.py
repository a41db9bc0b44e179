#!/usr/bin/env python3
"""Fixture for the CALLER of the hot path: the reference's own
``SihlLightningModule.configure_optimizers`` (src/sihl/lightning_module.py:179-245) and ``training_step`` (:68-120)
executed on a seeded model and batch.  Runs only in the build container (needs /root/reference).

``lightning`` is absent offline; ``configure_optimizers`` / ``training_step`` use ``pl.LightningModule`` only as an
``nn.Module`` with logging hooks, so a stand-in package provides exactly that (LightningModule = nn.Module + inert
``log`` / ``log_dict`` / ``lr_schedulers`` / ``optimizers``).  ``sihl/sihl_model.py`` runs unmodified; the heads /
neck are the reference's files as loaded by make_golden.load_reference (torchvision stand-ins as documented there);
the backbone is tests/golden/util.TinyBackbone (a plain torch.nn module: the grouping rule only looks at module types
and parameter names).

Stored: parameter name -> (lr, weight_decay) of the optimizer the reference builds; the learning-rate sequence of every
group over the first scheduler steps (LinearLR warm-up then the scheduler); the state_dict; the loss of one training
step, every parameter's gradient norm, and every parameter's sum after one optimizer step."""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from util import CALLER, TinyBackbone, caller_batch  # noqa: E402


def load_lightning_module(ns):
    class MisconfigurationException(Exception):
        pass

    class LightningModule(nn.Module):
        logger, global_step = None, 0

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        def lr_schedulers(self):
            return None

        def optimizers(self):
            raise AttributeError("no trainer attached")

    lightning = types.ModuleType("lightning")
    lightning.__path__ = []
    for name in ("lightning.fabric", "lightning.fabric.utilities"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    exc = types.ModuleType("lightning.fabric.utilities.exceptions")
    exc.MisconfigurationException = MisconfigurationException
    pl = types.ModuleType("lightning.pytorch")
    pl.LightningModule = LightningModule
    lightning.pytorch = pl
    sys.modules.update({"lightning": lightning, "lightning.fabric.utilities.exceptions": exc, "lightning.pytorch": pl})
    heads = sys.modules["sihl.heads"]
    heads.Head = type("Head", (), {})                                # typing.Protocol in the reference: annotations only
    heads.ViewInvarianceLearning = type("ViewInvarianceLearning", (), {})  # isinstance() test in training_step
    viz = types.ModuleType("sihl.visualization")
    viz.visualize = lambda **k: None
    sys.modules["sihl.visualization"] = viz
    mg._load("sihl.sihl_model", "sihl_model.py")
    return mg._load("sihl.lightning_module", "lightning_module.py")


def main():
    ns = mg.load_reference()
    lm_mod = load_lightning_module(ns)
    SihlModel = sys.modules["sihl.sihl_model"].SihlModel
    torch.manual_seed(4321)
    torch.set_num_threads(4)
    bb = TinyBackbone()
    neck = ns.BiFPN(bb.out_channels, CALLER["neck_channels"], CALLER["bottom"], CALLER["top"], num_layers=1)
    head = ns.ObjectDetection(neck.out_channels, CALLER["num_classes"], CALLER["bottom"], CALLER["top"],
                              num_channels=CALLER["neck_channels"])
    model = SihlModel(bb, neck, [head])
    lm = lm_mod.SihlLightningModule(model, optimizer=torch.optim.AdamW, optimizer_kwargs=dict(CALLER["opt"]),
                                    scheduler=torch.optim.lr_scheduler.CosineAnnealingLR,
                                    scheduler_kwargs={"T_max": CALLER["t_max"], "warmup": CALLER["warmup"]})
    lm.train()
    out = {f"sd.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
    cfg = lm.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    assert cfg["lr_scheduler"]["interval"] == "step"
    name_of = {id(p): n[len("model."):] for n, p in lm.named_parameters()}
    names, lrs, wds, gidx = [], [], [], []
    for gi, g in enumerate(opt.param_groups):
        for p in g["params"]:
            names.append(name_of[id(p)])
            lrs.append(g["initial_lr"] if "initial_lr" in g else g["lr"])
            wds.append(g["weight_decay"])
            gidx.append(gi)
    order = np.argsort(names)
    out["group.names"] = np.array(names)[order]
    out["group.base_lr"] = np.array(lrs, dtype=np.float64)[order]
    out["group.weight_decay"] = np.array(wds, dtype=np.float64)[order]
    gidx = np.array(gidx)[order]
    x, target = caller_batch()
    loss = lm.training_step((x, [target]), 1)
    loss.backward()
    out["step.loss"] = loss.detach().numpy()
    out["step.grad_norm"] = np.array([float(dict(model.named_parameters())[n].grad.norm()) for n in out["group.names"]])
    seq = []
    for _ in range(CALLER["sched_steps"]):
        seq.append([g["lr"] for g in opt.param_groups])
        opt.step()
        sched.step()
    seq = np.array(seq, dtype=np.float64)              # (steps, groups)
    out["sched.lr_by_param"] = seq[:, gidx]            # (steps, params): independent of the group order
    out["step.param_sum_after"] = np.array([float(dict(model.named_parameters())[n].detach().double().sum())
                                            for n in out["group.names"]])
    out["pinned_by"] = np.array("reference lightning_module.py + sihl_model.py (lightning stand-in) + tv-standins")
    path = os.path.join(HERE, "caller_step.npz")
    np.savez_compressed(path, **out)
    print(f"caller_step {os.path.getsize(path) / 1024:.1f} KiB, {len(names)} parameters, loss {float(loss):.6f}")


if __name__ == "__main__":
    main()
