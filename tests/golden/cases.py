"""Shared definitions of the golden-vector cases.

A case is (build, inputs, run):
  build(ns)            -> nn.Module built from the classes in namespace ``ns``
                          (the reference, the oracle, or sihl_amd - same ctor args)
  inputs()             -> dict name -> tensor / list of tensors (seeded, CPU fp32)
  run(module, inputs)  -> dict name -> tensor of results (outputs, input grads, param grads)

``tests/golden/make_golden.py`` runs every case on the REFERENCE's own files and
stores inputs, the module's state_dict (before and after the call, so BN running
statistics are pinned) and the results in ``tests/golden/<case>.npz``.  The tests
re-run the same case on the oracle (CPU) and on the HIP path (GPU) and compare.
"""
from typing import Callable, Dict, List

import torch
from torch import Tensor, nn


def _randn(seed: int, *shape) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def perturb_(module: nn.Module, seed: int, scale: float = 0.3) -> nn.Module:
    """Make weights 'trained-like' so nothing is pinned only at its init value
    (fusion weights == 1, LayerNorm gamma == 1, BN running stats == 0/1 ...)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.add_(scale * p.abs().mean().clamp(min=0.1) * torch.randn(p.shape, generator=g))
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.2 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    return module


def _levels(seed: int, batch: int, chans: List[int], size: int, used: range) -> List[Tensor]:
    """A level list; levels outside ``used`` are tiny placeholders except level 0, whose
    SHAPE carries the full image size (the heads read it)."""
    out = []
    for lvl, c in enumerate(chans):
        s = size // 2 ** lvl
        if lvl in used:
            out.append(_randn(seed + lvl, batch, c, s, s))
        elif lvl == 0:
            out.append(torch.zeros(batch, c, size, size))
        else:
            out.append(_randn(seed + lvl, batch, c, 2, 2))
    return out


def _grads(outs: List[Tensor], cots: List[Tensor], wrt: List[Tensor]) -> List[Tensor]:
    loss = sum((o * c.to(o.device, o.dtype)).sum() for o, c in zip(outs, cots))
    return torch.autograd.grad(loss, wrt, allow_unused=True)


def _with_param_grads(module: nn.Module, outs, cots, leaf_inputs: List[Tensor], res: Dict[str, Tensor]):
    params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
    g = _grads(outs, cots, leaf_inputs + [p for _, p in params])
    for i, gi in enumerate(g[: len(leaf_inputs)]):
        res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, g[len(leaf_inputs):]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


class Case:
    def __init__(self, name: str, build: Callable, inputs: Callable, run: Callable, train: bool,
                 needs: str = "layers"):
        self.name, self.build, self.inputs, self.run, self.train, self.needs = name, build, inputs, run, train, needs


CASES: Dict[str, Case] = {}


def _register(name, build, inputs, run, train, needs="layers"):
    CASES[name] = Case(name, build, inputs, run, train, needs)


# ------------------------------------------------------------------ single-tensor module cases
def _simple(name, build, shape, train, seed, grads=True):
    def inputs():
        return {"x": _randn(seed, *shape)}

    def run(m, inp):
        x = inp["x"].clone().requires_grad_(grads)
        y = m(x)
        res = {"y": y}
        if grads:
            _with_param_grads(m, [y], [_randn(seed + 100, *y.shape)], [x], res)
        return res

    _register(name, build, inputs, run, train)


_simple("cna3x3_train", lambda ns: perturb_(ns.ConvNormAct(32, 32), 1), (2, 32, 12, 16), True, 10)
_simple("cna3x3_eval", lambda ns: perturb_(ns.ConvNormAct(32, 64), 2), (2, 32, 8, 8), False, 11, grads=False)
_simple("cna1x1_train", lambda ns: perturb_(ns.ConvNormAct(64, 32, kernel_size=1), 3), (3, 64, 8, 8), True, 12)
_simple("cna1x1_eval", lambda ns: perturb_(ns.ConvNormAct(64, 32, kernel_size=1), 3), (3, 64, 8, 8), False, 12, grads=False)
_simple("cna3x3_silu_nonorm", lambda ns: perturb_(ns.ConvNormAct(32, 32, norm=None, act="silu"), 4), (2, 32, 8, 8), True, 13)
_simple("cna3x3_sigmoid_4to1", lambda ns: perturb_(ns.ConvNormAct(4, 1, norm=None, act="sigmoid"), 5), (2, 4, 8, 8), True, 14)
_simple("blurpool_s2", lambda ns: ns.BlurPool2d(32, stride=2), (2, 32, 8, 12), True, 15)
_simple("blurpool_s2_2x2", lambda ns: ns.BlurPool2d(32, stride=2), (2, 32, 2, 2), True, 16)
_simple("downscaler_train", lambda ns: perturb_(ns.AntialiasedDownscaler(32, 32), 6), (2, 32, 8, 8), True, 17)
_simple("interpolate_x2", lambda ns: ns.Interpolate(scale=2), (2, 32, 4, 6), True, 18)
_simple("interpolate_x2_1x1", lambda ns: ns.Interpolate(scale=2), (2, 32, 1, 1), True, 19)
_simple("upscaler_train", lambda ns: perturb_(ns.SimpleUpscaler(32, 32), 7), (2, 32, 4, 4), True, 20)


# ------------------------------------------------------------------ fusion
def _fusion(name, n, seed):
    def inputs():
        return {f"x{i}": _randn(seed + i, 2, 32, 6, 6) for i in range(n)}

    def run(m, inp):
        xs = [inp[f"x{i}"].clone().requires_grad_(True) for i in range(n)]
        y = m(xs)
        return _with_param_grads(m, [y], [_randn(seed + 100, *y.shape)], xs, {"y": y})

    _register(name, lambda ns: perturb_(ns.FastNormalizedFusion(n), seed), inputs, run, True)


_fusion("fusion2", 2, 30)
_fusion("fusion3", 3, 31)


# ------------------------------------------------------------------ level-list -> level-list necks
def _neck(name, build, chans, size, used, batch, train, seed, out_levels):
    def inputs():
        return {"levels": _levels(seed, batch, chans, size, used)}

    def run(m, inp):
        lv = [t.clone().requires_grad_(train and i in used) for i, t in enumerate(inp["levels"])]
        outs = m(lv)
        res = {"n_out": torch.tensor(len(outs))}
        sel = [outs[l] for l in out_levels]
        for l, o in zip(out_levels, sel):
            res[f"out{l}"] = o
        if train:
            cots = [_randn(seed + 200 + l, *o.shape) for l, o in zip(out_levels, sel)]
            _with_param_grads(m, sel, cots, [lv[i] for i in used], res)
        return res

    _register(name, build, inputs, run, train)


def _bifpn_layer_case(train):
    def inputs():
        return {"levels": [_randn(40 + i, 2, 32, 32 // 2 ** i, 32 // 2 ** i) for i in range(5)]}

    def run(m, inp):
        lv = [t.clone().requires_grad_(train) for t in inp["levels"]]
        outs = m(lv)
        res = {f"out{i}": o for i, o in enumerate(outs)}
        if train:
            cots = [_randn(240 + i, *o.shape) for i, o in enumerate(outs)]
            _with_param_grads(m, list(outs), cots, lv, res)
        return res

    _register("bifpn_layer_" + ("train" if train else "eval"),
              lambda ns: perturb_(ns.BiFPNLayer(32, 5), 8), inputs, run, train)


_bifpn_layer_case(True)
_bifpn_layer_case(False)

_BIFPN_CH = [3, 4, 8, 32, 64, 96]
_neck("bifpn_3to7_train", lambda ns: perturb_(ns.BiFPN(_BIFPN_CH, 32, 3, 7, num_layers=2), 9),
      _BIFPN_CH, 256, range(3, 6), 2, True, 50, [3, 4, 5, 6, 7])
_neck("bifpn_3to7_eval", lambda ns: perturb_(ns.BiFPN(_BIFPN_CH, 32, 3, 7, num_layers=2), 9),
      _BIFPN_CH, 256, range(3, 6), 2, False, 50, [3, 4, 5, 6, 7])
# pass-through configuration of tests/layers/test_bifpn.py:49-60 (levels 3-4 of a 6-level list)
_neck("bifpn_3to4_eval", lambda ns: perturb_(ns.BiFPN(_BIFPN_CH, 32, 3, 4), 10),
      _BIFPN_CH, 128, range(3, 6), 2, False, 60, [3, 4, 5])
_neck("fpn_3to5_train", lambda ns: perturb_(ns.FPN(list(_BIFPN_CH), 32, 3, 5), 11),
      _BIFPN_CH, 128, range(3, 6), 2, True, 70, [3, 4, 5], )
_neck("fpn_3to5_eval", lambda ns: perturb_(ns.FPN(list(_BIFPN_CH), 32, 3, 5), 11),
      _BIFPN_CH, 128, range(3, 6), 2, False, 70, [3, 4, 5])
_neck("fpn_3to7_eval", lambda ns: perturb_(ns.FPN(list(_BIFPN_CH), 32, 3, 7), 12),
      _BIFPN_CH, 256, range(3, 6), 2, False, 80, [3, 4, 5, 6, 7])
_neck("fpn_3to7_train", lambda ns: perturb_(ns.FPN(list(_BIFPN_CH), 32, 3, 7), 12),
      _BIFPN_CH, 256, range(3, 6), 2, True, 80, [3, 4, 5, 6, 7])
for c in ("fpn_3to5_train", "fpn_3to5_eval", "fpn_3to7_eval", "fpn_3to7_train"):
    CASES[c].needs = "fpn"


# ------------------------------------------------------------------ object-detection head
_OD_CH = [3] + [32] * 7


def _od_build(ns):
    return perturb_(ns.ObjectDetection(_OD_CH, num_classes=8, bottom_level=3, top_level=7,
                                       num_channels=32, num_layers=4), 13, scale=0.6)


def _od_inputs(seed=90):
    return {"levels": _levels(seed, 3, _OD_CH, 128, range(3, 8))}


def _od_forward_run(m, inp):
    with torch.no_grad():
        n, scores, classes, boxes = m(inp["levels"])
        off, scl = m.get_offsets_and_scales(inp["levels"])
    return {"num_instances": n, "scores": scores, "classes": classes, "boxes": boxes,
            "offsets": off, "scales": scl}


_register("od_forward_eval", _od_build, _od_inputs, _od_forward_run, False, needs="od")


def od_targets():
    """3 images: several objects, ZERO objects (tests/heads/test_object_detection.py:42), one object."""
    boxes = [torch.tensor([[8.0, 12.0, 60.0, 70.0], [40.0, 30.0, 120.0, 100.0], [64.0, 64.0, 96.0, 112.0]]),
             torch.zeros(0, 4), torch.tensor([[10.0, 20.0, 50.0, 90.0]])]
    classes = [torch.tensor([1, 5, 7]), torch.zeros(0, dtype=torch.int64), torch.tensor([3])]
    return classes, boxes


def _od_train_run(m, inp):
    lv = [t.clone().requires_grad_(i >= 3) for i, t in enumerate(inp["levels"])]
    dev = lv[3].device
    classes, boxes = od_targets()
    classes, boxes = [c.to(dev) for c in classes], [b.to(dev) for b in boxes]
    loss, metrics = m.training_step(lv, classes, boxes)
    res = {"loss": loss, **{k: v for k, v in metrics.items()}}
    params = [(n, p) for n, p in m.named_parameters()]
    g = torch.autograd.grad(loss, lv[3:] + [p for _, p in params], allow_unused=True)
    for i, gi in enumerate(g[:5]):
        res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, g[5:]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


_register("od_training_step", _od_build, _od_inputs, _od_train_run, True, needs="od")


def _od_nogt_run(m, inp):
    lv = [t.clone() for t in inp["levels"]]
    dev = lv[3].device
    loss, metrics = m.training_step(lv, [torch.zeros(0, dtype=torch.int64, device=dev)] * 3,
                                    [torch.zeros(0, 4, device=dev)] * 3)
    return {"loss": loss, **metrics}


_register("od_training_step_no_gt", _od_build, _od_inputs, _od_nogt_run, True, needs="od")


def _matching_inputs():
    return {"levels": _od_inputs()["levels"]}


def _matching_run(m, inp):
    off, scl = m.get_offsets_and_scales(inp["levels"])
    anchors = (off + scl) * torch.tensor([[128, 128, 128, 128]], device=off.device)
    res = {"anchors": anchors}
    _, boxes = od_targets()
    for b, gt in enumerate(boxes):
        gt = gt.to(anchors.device)
        a, r = m.bbox_matching(anchors, gt, 9, relative=True)
        a2, i2 = m.bbox_matching(anchors, gt, 9, relative=False)
        res[f"assign{b}"], res[f"rel_iou{b}"], res[f"iou{b}"] = a, r, i2
    return res


_register("od_bbox_matching", _od_build, _matching_inputs, _matching_run, False, needs="od")


# ------------------------------------------------------------------ semantic-segmentation head
_SS_CH = [3] + [32] * 5


def _ss_build(layers):
    return lambda ns: perturb_(ns.SemanticSegmentation(_SS_CH, num_classes=7, num_channels=32,
                                                       num_layers=layers), 14)


def _ss_inputs():
    return {"levels": _levels(110, 2, _SS_CH, 64, range(3, 6))}


def _ss_forward_run(m, inp):
    with torch.no_grad():
        logits = m.get_logits(inp["levels"])
        scores, classes = m(inp["levels"])
    return {"logits": logits, "scores": scores, "classes": classes}


def _ss_train_run(m, inp):
    lv = [t.clone().requires_grad_(i >= 3) for i, t in enumerate(inp["levels"])]
    g = torch.Generator().manual_seed(111)
    targets = torch.randint(0, 7, (2, 64, 64), generator=g).to(lv[3].device)
    loss, _ = m.training_step(lv, targets)
    res = {"loss": loss}
    params = [(n, p) for n, p in m.named_parameters()]
    gr = torch.autograd.grad(loss, lv[3:] + [p for _, p in params], allow_unused=True)
    for i, gi in enumerate(gr[:3]):
        res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, gr[3:]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


_register("semseg_forward_eval", _ss_build(2), _ss_inputs, _ss_forward_run, False, needs="semseg")
_register("semseg_forward_eval_nl0", _ss_build(0), _ss_inputs, _ss_forward_run, False, needs="semseg")
_register("semseg_training_step", _ss_build(2), _ss_inputs, _ss_train_run, True, needs="semseg")


def _sppm_case():
    def inputs():
        return {"x": _randn(120, 2, 32, 4, 4)}

    def run(m, inp):
        x = inp["x"].clone().requires_grad_(True)
        y = m(x)
        return _with_param_grads(m, [y], [_randn(320, *y.shape)], [x], {"y": y})

    _register("sppm_train", lambda ns: perturb_(ns.SPPM(32, 32), 15), inputs, run, True, needs="semseg")


def _uafm_case():
    def inputs():
        return {"x1": _randn(121, 2, 32, 8, 8), "x2": _randn(122, 2, 32, 8, 8)}

    def run(m, inp):
        x1 = inp["x1"].clone().requires_grad_(True)
        x2 = inp["x2"].clone().requires_grad_(True)
        y = m(x1, x2)
        return _with_param_grads(m, [y], [_randn(321, *y.shape)], [x1, x2], {"y": y})

    _register("uafm_train", lambda ns: perturb_(ns.UAFM(32, 32), 16), inputs, run, True, needs="semseg")


_sppm_case()
_uafm_case()


# ------------------------------------------------------------------ instance-segmentation head
_IS_CH = [3] + [32] * 5


def _is_build(ns):
    return perturb_(ns.InstanceSegmentation(_IS_CH, num_classes=6, mask_level=3, bottom_level=3, top_level=5,
                                            num_channels=32, num_layers=2, max_instances=12), 21, scale=0.6)


def _is_inputs(seed=95):
    return {"levels": _levels(seed, 2, _IS_CH, 64, range(3, 6))}


def _is_forward_run(m, inp):
    with torch.no_grad():
        n, scores, classes, masks = m(inp["levels"])
    return {"num_instances": n, "scores": scores, "classes": classes, "masks": masks}


_register("iseg_forward_eval", _is_build, _is_inputs, _is_forward_run, False, needs="iseg")


def iseg_targets():
    """2 images: two objects (one of them with an EMPTY mask, dropped by the head) and one object."""
    m0 = torch.zeros(3, 64, 64)
    m0[0, 8:30, 10:40] = 1.0
    m0[1, 30:60, 24:56] = 1.0
    m1 = torch.zeros(1, 64, 64)
    m1[0, 4:44, 12:36] = 1.0
    return [torch.tensor([2, 4, 1]), torch.tensor([5])], [m0, m1]


def _is_train_run(m, inp):
    lv = [t.clone().requires_grad_(i >= 3) for i, t in enumerate(inp["levels"])]
    dev = lv[3].device
    classes, masks = iseg_targets()
    loss, metrics = m.training_step(lv, [c.to(dev) for c in classes], [k.to(dev) for k in masks])
    res = {"loss": loss, **{k: v for k, v in metrics.items()}}
    params = [(n, p) for n, p in m.named_parameters()]
    g = torch.autograd.grad(loss, lv[3:] + [p for _, p in params], allow_unused=True)
    for i, gi in enumerate(g[:3]):
        res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, g[3:]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


_register("iseg_training_step", _is_build, _is_inputs, _is_train_run, True, needs="iseg")


# ------------------------------------------------------------------ depth-estimation head
def _depth_build(ns):
    return perturb_(ns.DepthEstimation(_SS_CH, lower_bound=0.5, upper_bound=10.0, bottom_level=3, top_level=5,
                                       num_channels=32, num_layers=1, num_bins=16), 23, scale=0.5)


def _depth_inputs():
    return _ss_inputs()


def _depth_forward_run(m, inp):
    with torch.no_grad():
        return {"depth": m(inp["levels"]), "bin_centers": m.get_bin_centers(inp["levels"])}


_register("depth_forward_eval", _depth_build, _depth_inputs, _depth_forward_run, False, needs="depth")


def _depth_train_run(m, inp):
    lv = [t.clone().requires_grad_(i >= 3) for i, t in enumerate(inp["levels"])]
    dev = lv[3].device
    H, W = lv[0].shape[2:]
    g = torch.Generator().manual_seed(77)
    targets = (0.5 + 9.5 * torch.rand(lv[0].shape[0], H, W, generator=g)).to(dev)
    masks = (torch.rand(lv[0].shape[0], H, W, generator=g) > 0.3).to(dev)
    loss, metrics = m.training_step(lv, targets, masks)
    res = {"loss": loss, **metrics}
    params = [(n, p) for n, p in m.named_parameters()]
    grads = torch.autograd.grad(loss, lv[3:] + [p for _, p in params], allow_unused=True)
    for i, gi in enumerate(grads[:len(lv) - 3]):
        res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, grads[len(lv) - 3:]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


_register("depth_training_step", _depth_build, _depth_inputs, _depth_train_run, True, needs="depth")


# ------------------------------------------------------------------ HybridEncoder neck (the examples' default neck)
_HE_CH = [3, 8, 16, 24, 40, 48]
_neck("hybrid_3to6_eval", lambda ns: perturb_(ns.HybridEncoder(list(_HE_CH), 32, 3, 6), 31, scale=0.3),
      _HE_CH, 128, range(3, 6), 2, False, 120, [3, 4, 5, 6])
_neck("hybrid_3to6_train", lambda ns: perturb_(ns.HybridEncoder(list(_HE_CH), 32, 3, 6), 31, scale=0.3),
      _HE_CH, 128, range(3, 6), 2, True, 121, [3, 4, 5, 6])
for c in ("hybrid_3to6_eval", "hybrid_3to6_train"):
    CASES[c].needs = "hybrid"


# ------------------------------------------------------------------ keypoint-detection head
def _kp_build(ns):
    return perturb_(ns.KeypointDetection(_IS_CH, num_keypoints=5, mask_level=3, bottom_level=4, top_level=5,
                                         num_channels=32, num_layers=2, max_instances=6), 27, scale=0.6)


def _kp_forward_run(m, inp):
    with torch.no_grad():
        n, scores, presence, kpts = m(inp["levels"])
        heat = m(inp["levels"], output_heatmaps=True)
    return {"num_instances": n, "scores": scores, "presence": presence, "keypoints": kpts, "heatmaps": heat}


_register("kpt_forward_eval", _kp_build, _is_inputs, _kp_forward_run, False, needs="kpt")


def kpt_targets():
    """2 images: two persons (one of them without any visible keypoint, dropped by the head) and one person."""
    k0 = torch.tensor([[[10.0, 12], [30, 14], [22, 40], [12, 50], [34, 52]],
                       [[1.0, 1], [2, 2], [3, 3], [4, 4], [5, 5]],
                       [[40.0, 8], [58, 10], [50, 30], [42, 44], [60, 46]]])
    p0 = torch.tensor([[True, True, False, True, True], [False] * 5, [True, True, True, True, False]])
    k1 = torch.tensor([[[20.0, 20], [44, 22], [32, 36], [24, 56], [46, 58]]])
    p1 = torch.tensor([[True, False, True, True, True]])
    return [p0, p1], [k0, k1]


def _kp_train_run(m, inp):
    lv = [t.clone().requires_grad_(i >= 3) for i, t in enumerate(inp["levels"])]
    dev = lv[3].device
    presence, keypoints = kpt_targets()
    loss, metrics = m.training_step(lv, [p.to(dev) for p in presence], [k.to(dev) for k in keypoints])
    res = {"loss": loss, **{k: v for k, v in metrics.items()}}
    params = [(n, p) for n, p in m.named_parameters()]
    g = torch.autograd.grad(loss, lv[3:] + [p for _, p in params], allow_unused=True)
    for i, gi in enumerate(g[:3]):
        if gi is not None:
            res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, g[3:]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


_register("kpt_training_step", _kp_build, _is_inputs, _kp_train_run, True, needs="kpt")


# ------------------------------------------------------------------ quadrilateral-detection head
def _quad_build(ns):
    return perturb_(ns.QuadrilateralDetection(_IS_CH, num_classes=4, bottom_level=3, top_level=5, num_channels=32,
                                              num_layers=2, max_instances=8), 29, scale=0.6)


def _quad_forward_run(m, inp):
    with torch.no_grad():
        n, scores, classes, quads = m(inp["levels"])
    return {"num_instances": n, "scores": scores, "classes": classes, "quads": quads}


_register("quad_forward_eval", _quad_build, _is_inputs, _quad_forward_run, False, needs="quad")


def quad_targets():
    """2 images: two quadrilaterals (one given with a concave vertex order) and one."""
    q0 = torch.tensor([[[8.0, 10], [40, 6], [46, 36], [12, 42]], [[30.0, 30], [58, 34], [44, 44], [56, 60]]])
    q1 = torch.tensor([[[20.0, 8], [52, 14], [48, 50], [16, 44]]])
    return [torch.tensor([1, 3]), torch.tensor([0])], [q0, q1]


def _quad_train_run(m, inp):
    lv = [t.clone().requires_grad_(i >= 3) for i, t in enumerate(inp["levels"])]
    dev = lv[3].device
    classes, quads = quad_targets()
    loss, metrics = m.training_step(lv, [c.to(dev) for c in classes], [q.to(dev) for q in quads])
    res = {"loss": loss, **{k: v for k, v in metrics.items()}}
    params = [(n, p) for n, p in m.named_parameters()]
    g = torch.autograd.grad(loss, lv[3:] + [p for _, p in params], allow_unused=True)
    for i, gi in enumerate(g[:3]):
        if gi is not None:
            res[f"gin{i}"] = gi
    for (n, _), gp in zip(params, g[3:]):
        if gp is not None:
            res[f"gp.{n}"] = gp
    return res


_register("quad_training_step", _quad_build, _is_inputs, _quad_train_run, True, needs="quad")
