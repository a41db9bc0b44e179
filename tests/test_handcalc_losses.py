"""CPU: CIoU / CIoU-loss / sigmoid-focal-loss arithmetic pinned by numbers that do NOT come from `oracle`.

The fixture generator's torchvision stand-ins for these three functions are the oracle's own (torchvision 0.21.0 is not
in the image), so the reference-generated fixtures cannot catch a mistake in them.  Here every expected
value is worked out from the published definitions with exact fractions, `math.atan` and `math.log` on hand-picked
boxes, plus a scalar pure-Python restatement (no tensors, no shared code) for a finite-difference check of the
gradient semantics (alpha is a constant: torchvision computes it under no_grad).

    CIoU(b, g)      = IoU - rho^2 / c^2 - alpha * v
    v               = (4 / pi^2) * (atan(w_g / h_g) - atan(w / h))^2
    alpha           = v / (1 - IoU + v + eps)                           (no gradient)
    CIoU-loss(b, g) = 1 - IoU + rho^2 / c^2 + alpha * v,  IoU = inter / (union + eps)
    focal(x, t)     = alpha_t * (1 - p_t)^gamma * BCE(x, t),  p = sigmoid(x), p_t = p t + (1-p)(1-t),
                      alpha_t = alpha t + (1 - alpha)(1 - t);  alpha = 0.25, gamma = 2
"""
import math

import pytest
import torch

import oracle.heads as oh
from sihl_amd.heads import box_ops

EPS = 1e-7
IMPLS = [("oracle", oh.complete_box_iou, oh.complete_box_iou_loss),
         ("sihl_amd", box_ops.complete_box_iou, box_ops.complete_box_iou_loss)]


def _v(w1, h1, w2, h2):
    return 4 / math.pi ** 2 * (math.atan(w1 / h1) - math.atan(w2 / h2)) ** 2


# (box, ground truth, IoU as an exact fraction, rho^2, c^2, v) - worked by hand:
#  A  identical unit boxes: everything vanishes, CIoU = 1
#  B  (0,0,2,2) vs (1,1,3,3): inter 1, union 4+4-1 = 7; centres (1,1),(2,2): rho^2 = 2; hull (0,0,3,3): c^2 = 18;
#     both square: v = 0                                              -> CIoU = 1/7 - 1/9
#  C  (0,0,4,2) vs (0,0,2,4): inter 2x2 = 4, union 8+8-4 = 12; centres (2,1),(1,2): rho^2 = 2; hull 4x4: c^2 = 32;
#     aspect 2 vs 1/2: v = (4/pi^2)(atan 2 - atan 1/2)^2
#  D  disjoint (0,0,1,1) vs (2,0,3,2): inter 0, union 1+2 = 3; centres (.5,.5),(2.5,1): rho^2 = 4.25; hull 3x2: c^2 = 13;
#     aspect 1 vs 1/2: v = (4/pi^2)(atan 1 - atan 1/2)^2
#  E  containment (0,0,4,4) vs (1,1,3,2): inter 2, union 16; centres (2,2),(2,1.5): rho^2 = .25; hull 4x4: c^2 = 32;
#     aspect 1 vs 2: v = (4/pi^2)(atan 1 - atan 2)^2
CASES = [
    ("A", (0, 0, 1, 1), (0, 0, 1, 1), 1.0, 0.0, 2.0, 0.0),
    ("B", (0, 0, 2, 2), (1, 1, 3, 3), 1 / 7, 2.0, 18.0, 0.0),
    ("C", (0, 0, 4, 2), (0, 0, 2, 4), 4 / 12, 2.0, 32.0, _v(4, 2, 2, 4)),
    ("D", (0, 0, 1, 1), (2, 0, 3, 2), 0.0, 4.25, 13.0, _v(1, 1, 1, 2)),
    ("E", (0, 0, 4, 4), (1, 1, 3, 2), 2 / 16, 0.25, 32.0, _v(4, 4, 2, 1)),
]


def _expected_ciou(iou, rho2, c2, v):
    alpha = v / (1 - iou + v + EPS)
    return iou - rho2 / (c2 + EPS) - alpha * v


@pytest.mark.parametrize("name,ciou,loss", IMPLS)
def test_ciou_matrix_hand_values(name, ciou, loss):
    b = torch.tensor([c[1] for c in CASES], dtype=torch.float64)
    g = torch.tensor([c[2] for c in CASES], dtype=torch.float64)
    got = ciou(b, g)  # (N, N): the hand cases sit on the diagonal
    for i, (tag, _, _, iou, rho2, c2, v) in enumerate(CASES):
        assert float(got[i, i]) == pytest.approx(_expected_ciou(iou, rho2, c2, v), abs=1e-9), (name, tag)
    # two literal numbers, so that the helper above is not the only witness
    assert float(got[1, 1]) == pytest.approx(1 / 7 - 1 / 9, abs=1e-7)
    assert float(got[0, 0]) == pytest.approx(1.0, abs=1e-6)
    # symmetric in its arguments up to the role of alpha (v is symmetric, IoU / rho / c are)
    assert torch.allclose(got, ciou(g, b).T, atol=1e-12)


@pytest.mark.parametrize("name,ciou,loss", IMPLS)
def test_ciou_loss_hand_values(name, ciou, loss):
    b = torch.tensor([c[1] for c in CASES], dtype=torch.float64)
    g = torch.tensor([c[2] for c in CASES], dtype=torch.float64)
    got = loss(b, g)
    union = [1.0, 7.0, 12.0, 3.0, 16.0]
    for i, (tag, _, _, iou, rho2, c2, v) in enumerate(CASES):
        iou_eps = iou * union[i] / (union[i] + EPS)  # the loss divides by union + eps
        alpha = v / (1 - iou_eps + v + EPS)
        want = 1 - iou_eps + rho2 / (c2 + EPS) + alpha * v
        assert float(got[i]) == pytest.approx(want, abs=1e-9), (name, tag)
    assert float(got[1]) == pytest.approx(1 - 1 / 7 + 1 / 9, abs=1e-6)
    assert float(got[3]) == pytest.approx(1 + 4.25 / 13 + _v(1, 1, 1, 2) ** 2 / (1 + _v(1, 1, 1, 2)), abs=1e-6)


def _scalar_loss(b, g, alpha_fixed=None):
    """CIoU loss of one box pair in plain Python floats (independent of both tensor implementations)."""
    x1, y1, x2, y2 = b
    X1, Y1, X2, Y2 = g
    iw, ih = min(x2, X2) - max(x1, X1), min(y2, Y2) - max(y1, Y1)
    inter = iw * ih if iw > 0 and ih > 0 else 0.0
    union = (x2 - x1) * (y2 - y1) + (X2 - X1) * (Y2 - Y1) - inter
    iou = inter / (union + EPS)
    c2 = (max(x2, X2) - min(x1, X1)) ** 2 + (max(y2, Y2) - min(y1, Y1)) ** 2 + EPS
    rho2 = ((x1 + x2 - X1 - X2) / 2) ** 2 + ((y1 + y2 - Y1 - Y2) / 2) ** 2
    v = 4 / math.pi ** 2 * (math.atan((X2 - X1) / (Y2 - Y1)) - math.atan((x2 - x1) / (y2 - y1))) ** 2
    alpha = v / (1 - iou + v + EPS) if alpha_fixed is None else alpha_fixed
    return 1 - iou + rho2 / c2 + alpha * v, alpha


@pytest.mark.parametrize("name,ciou,loss", IMPLS)
def test_ciou_loss_gradient_treats_alpha_as_constant(name, ciou, loss):
    b0, g0 = (0.3, 0.1, 3.7, 2.2), (1.0, 0.6, 2.9, 3.1)
    b = torch.tensor([b0], dtype=torch.float64, requires_grad=True)
    g = torch.tensor([g0], dtype=torch.float64)
    out = loss(b, g)
    want, alpha = _scalar_loss(b0, g0)
    assert float(out) == pytest.approx(want, abs=1e-12)
    out.sum().backward()
    h = 1e-6
    for k in range(4):
        hi = list(b0); hi[k] += h
        lo = list(b0); lo[k] -= h
        fd = (_scalar_loss(hi, g0, alpha)[0] - _scalar_loss(lo, g0, alpha)[0]) / (2 * h)  # alpha held fixed
        assert float(b.grad[0, k]) == pytest.approx(fd, abs=1e-6), (name, k)
    # with alpha differentiated the gradient would differ visibly on this pair: the test has teeth
    k = 2
    hi = list(b0); hi[k] += h
    lo = list(b0); lo[k] -= h
    fd_full = (_scalar_loss(hi, g0)[0] - _scalar_loss(lo, g0)[0]) / (2 * h)
    assert abs(fd_full - float(b.grad[0, k])) > 1e-4


def test_sigmoid_focal_loss_hand_values():
    ln2 = math.log(2.0)
    s1 = 1 / (1 + math.exp(-1.0))  # sigmoid(1)
    cases = [
        # x, t, expected
        (0.0, 1.0, 0.25 * 0.5 ** 2 * ln2),                      # p = 1/2: alpha_t 1/4, (1-p_t)^2 1/4, BCE ln 2
        (0.0, 0.0, 0.75 * 0.5 ** 2 * ln2),                      # negative class: alpha_t 3/4
        (1.0, 1.0, 0.25 * (1 - s1) ** 2 * -math.log(s1)),
        (1.0, 0.0, 0.75 * s1 ** 2 * -math.log(1 - s1)),
        (-30.0, 0.0, 0.75 * (1 / (1 + math.exp(30.0))) ** 2 * math.log1p(math.exp(-30.0))),  # easy negative: ~0
    ]
    x = torch.tensor([c[0] for c in cases], dtype=torch.float64)
    t = torch.tensor([c[1] for c in cases], dtype=torch.float64)
    got = oh.sigmoid_focal_loss(x, t)
    for i, (_, _, want) in enumerate(cases):
        assert float(got[i]) == pytest.approx(want, rel=1e-9, abs=1e-30)
    assert float(got[0]) == pytest.approx(0.043321698784996585, abs=1e-12)  # ln 2 / 16
