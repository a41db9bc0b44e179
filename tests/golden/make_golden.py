#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own source files.

Runs ONLY in the build container (needs /root/reference); nothing here travels
to the GPU box except the .npz data it writes.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

How the reference is loaded: ``sihl/__init__.py`` cannot be
imported (torchvision / torchmetrics / timm / lightning are absent), so bare
package objects are registered for ``sihl``, ``sihl.layers`` and ``sihl.heads``
and each hot-path file is executed with importlib from where it lies:
  layers/{convblocks,pooling,scalers,bifpn}.py  - need only torch/einops/numpy: run UNMODIFIED
  layers/fpn.py, layers/hybrid_encoder.py, utils/__init__.py (+ polygon_iou, pck, f1), heads/object_detection.py,
  heads/semantic_segmentation.py, heads/instance_segmentation.py, heads/depth_estimation.py,
  heads/keypoint_detection.py, heads/quadrilateral_detection.py
      - additionally import ``torchvision.ops`` / ``torchmetrics``.  Stand-in modules are
        registered for those imports: ``ops.Conv2dNormActivation`` / ``ops.MLP`` (compositions
        of torch.nn layers, documented structure) and ``ops.complete_box_iou[_loss]`` (published
        CIoU definition) come from ``oracle``; torchmetrics names are inert placeholders
        (validation path only).  Fixtures whose ``pinned_by`` says "reference+tv-standins"
        therefore pin the reference's OWN code (flattening, top-k, matching, losses, decode)
        but not torchvision's arithmetic.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = "/root/reference/src/sihl"


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    import oracle.heads as oh
    import oracle.layers as ol

    sihl = _pkg("sihl", REF)
    layers = _pkg("sihl.layers", REF + "/layers")
    heads = _pkg("sihl.heads", REF + "/heads")
    sihl.layers, sihl.heads = layers, heads

    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    ops.Conv2dNormActivation, ops.MLP = ol.Conv2dNormActivation, ol.MLP
    ops.complete_box_iou = oh.complete_box_iou
    ops.complete_box_iou_loss = lambda a, b, reduction="none", eps=1e-7: oh.complete_box_iou_loss(a, b, eps)
    ops.masks_to_boxes = oh.masks_to_boxes
    ops.sigmoid_focal_loss = lambda x, t, alpha=0.25, gamma=2.0, reduction="none": oh.sigmoid_focal_loss(x, t, alpha, gamma)
    tv.ops = ops
    sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, ops
    tm = types.ModuleType("torchmetrics")
    for n in ("MeanMetric", "JaccardIndex", "Accuracy", "Precision", "Recall"):
        setattr(tm, n, type(n, (), {}))
    tmd = types.ModuleType("torchmetrics.detection")
    tmm = types.ModuleType("torchmetrics.detection.mean_ap")
    tmm.MeanAveragePrecision = type("MeanAveragePrecision", (), {})
    sys.modules.update({"torchmetrics": tm, "torchmetrics.detection": tmd,
                        "torchmetrics.detection.mean_ap": tmm})

    ns = types.SimpleNamespace()
    LAYER_FILES = ("convblocks", "pooling", "scalers", "bifpn", "fpn")
    for name in LAYER_FILES:
        mod = _load(f"sihl.layers.{name}", f"layers/{name}.py")
        for k, v in vars(mod).items():
            if isinstance(v, type) and v.__module__ == mod.__name__:
                setattr(layers, k, v)
                setattr(ns, k, v)
    # sihl/utils/__init__.py itself runs unmodified once torchmetrics.Metric / torchvision.ops.box_iou resolve (inert
    # placeholders: only the metric classes use them); it provides EPS and sine_embedding_2d_grid
    tm.Metric = type("Metric", (), {})
    ops.box_iou = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("placeholder"))
    tv.__path__ = []
    _pkg("sihl.utils", REF + "/utils")
    for sub in ("polygon_iou", "pck", "f1"):
        _load(f"sihl.utils.{sub}", f"utils/{sub}.py")
    utils = _load("sihl.utils", "utils/__init__.py")
    sihl.utils = utils
    mod = _load("sihl.layers.hybrid_encoder", "layers/hybrid_encoder.py")  # needs sihl.utils + tv stand-ins
    for k, v in vars(mod).items():
        if isinstance(v, type) and v.__module__ == mod.__name__:
            setattr(layers, k, v)
            setattr(ns, k, v)
    tmr = types.ModuleType("torchmetrics.regression")
    tmr.MeanAbsoluteError, tmr.MeanSquaredError = type("MeanAbsoluteError", (), {}), type("MeanSquaredError", (), {})
    sys.modules["torchmetrics.regression"] = tmr
    for name in ("object_detection", "semantic_segmentation", "instance_segmentation", "depth_estimation",
                 "keypoint_detection", "quadrilateral_detection"):
        mod = _load(f"sihl.heads.{name}", f"heads/{name}.py")
        for k, v in vars(mod).items():
            if isinstance(v, type) and v.__module__ == mod.__name__:
                setattr(heads, k, v)
                setattr(ns, k, v)
    return ns


def _flatten(prefix, obj, out):
    if isinstance(obj, (list, tuple)):
        out[f"{prefix}.len"] = np.array(len(obj))
        for i, t in enumerate(obj):
            _flatten(f"{prefix}.{i}", t, out)
    else:
        out[prefix] = obj.detach().cpu().numpy()


PINNED = {"layers": "reference-unmodified", "fpn": "reference+tv-standins",
          "od": "reference+tv-standins", "semseg": "reference+tv-standins", "iseg": "reference+tv-standins", "depth": "reference+tv-standins", "hybrid": "reference+tv-standins", "kpt": "reference+tv-standins", "quad": "reference+tv-standins"}


def main(argv):
    from cases import CASES

    ns = load_reference()
    torch.set_num_threads(4)
    names = argv or list(CASES)
    for name in names:
        case = CASES[name]
        torch.manual_seed(1234)
        m = case.build(ns)
        m.train(case.train)
        out = {"pinned_by": np.array(PINNED[case.needs])}
        for k, v in m.state_dict().items():
            out[f"sd.{k}"] = v.detach().cpu().numpy().copy()
        inp = case.inputs()
        for k, v in inp.items():
            _flatten(f"in.{k}", v, out)
        res = case.run(m, inp)
        for k, v in res.items():
            out[f"res.{k}"] = v.detach().cpu().numpy()
        for k, v in m.state_dict().items():  # BN running stats after the call
            if "running_" in k or "num_batches" in k:
                out[f"sd_after.{k}"] = v.detach().cpu().numpy().copy()
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB  {len(res)} results")


if __name__ == "__main__":
    main(sys.argv[1:])
