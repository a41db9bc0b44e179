# Same public names as the reference's ``sihl.heads`` for the hot-path heads (src/sihl/heads/__init__.py).
from sihl_amd.heads.depth_estimation import DepthEstimation  # noqa: F401
from sihl_amd.heads.instance_segmentation import InstanceSegmentation  # noqa: F401
from sihl_amd.heads.keypoint_detection import KeypointDetection  # noqa: F401
from sihl_amd.heads.mlp import MLP  # noqa: F401
from sihl_amd.heads.multiclass_classification import MulticlassClassification  # noqa: F401
from sihl_amd.heads.object_detection import ObjectDetection  # noqa: F401
from sihl_amd.heads.quadrilateral_detection import QuadrilateralDetection  # noqa: F401
from sihl_amd.heads.semantic_segmentation import SPPM, UAFM, SemanticSegmentation  # noqa: F401
