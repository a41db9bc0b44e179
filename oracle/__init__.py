"""CPU oracle for the sihl hot path (TEST INFRASTRUCTURE — not product code).

This package is a pure-``torch`` CPU restatement of the reference's
backbone -> FPN/BiFPN -> dense-head path.  It exists so that the HIP path in
``sihl_amd`` can be checked on a GPU box where ``/root/reference`` does not
exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; ``sihl_amd`` never does.

Pinning: the restatement is compared (CPU vs CPU, <=1e-6) against golden
vectors generated in the build container from the reference's own files
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).  Pieces whose
arithmetic lives in torchvision 0.21 (absent from the container: CIoU
matching/loss, ResNet50) are restated from the published definitions and are
"parity unpinned" - see DESIGN.md.
"""
from oracle.layers import (  # noqa: F401
    ConvNormAct,
    SequentialConvBlocks,
    BlurPool2d,
    AntialiasedDownscaler,
    Interpolate,
    SimpleUpscaler,
    FastNormalizedFusion,
    BiFPNLayer,
    BiFPN,
    FPN,
    Conv2dNormActivation,
    MLP,
)
from oracle.heads import (  # noqa: F401
    ObjectDetection,
    QuadrilateralDetection,
    KeypointDetection,
    InstanceSegmentation,
    DepthEstimation,
    SemanticSegmentation,
    SPPM,
    UAFM,
    MulticlassClassification,
    complete_box_iou,
    complete_box_iou_loss,
)
from oracle.model import SihlModel, ResNetBackbone, TimmBackbone  # noqa: F401
